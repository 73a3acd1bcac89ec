"""Batch collation for the T5 path: text -> the 5 (or 6) int64 tensors the model consumes.

Mirrors /root/reference/src/src_t5/processor/Collator.py (`Collator.__call__` :8-34, `TestCollator` :36-68,
`calculate_whole_word_ids` :72-83): same inputs (list of {'input','output'[,'user_idx']}), same outputs and order
(input_ids, attention_mask, whole_word_ids, output_ids, output_attention[, user_idx]), padding "longest", truncation at
512.  The whole-word index is computed for the whole batch at once from a per-vocabulary "piece starts with U+2581"
table instead of a Python loop over every token; `calculate_whole_word_ids` keeps the
reference's per-sequence signature and its quirk that only the LAST column is zeroed.
"""
from typing import List, Sequence

import numpy as np
import torch

WORD_START = "▁"


def calculate_whole_word_ids(tokenized_text: Sequence[str], input_ids: Sequence[int]) -> List[int]:
    out, curr = [], 0
    for piece in tokenized_text:
        if piece == "<pad>":
            curr = 0
        if piece.startswith(WORD_START):
            curr += 1
        out.append(curr)
    return out[: len(input_ids) - 1] + [0]


def _encode(tokenizer, texts):
    enc = tokenizer(list(texts), padding="longest", truncation=True, max_length=512)
    return enc["input_ids"], enc["attention_mask"]


class Collator:
    def __init__(self, tokenizer):
        self.tokenizer = tokenizer
        self._starts = None
        self._pad_id = getattr(tokenizer, "pad_token_id", 0) or 0

    def _start_table(self, max_id: int) -> np.ndarray:
        if self._starts is None or len(self._starts) <= max_id:
            n = max(len(self.tokenizer), max_id + 1)
            pieces = self.tokenizer.convert_ids_to_tokens(list(range(n)))
            self._starts = np.array([(p is not None and p.startswith(WORD_START)) for p in pieces], dtype=np.int64)
        return self._starts

    def whole_word_ids(self, input_ids: np.ndarray) -> np.ndarray:
        """Vectorised `calculate_whole_word_ids` over a padded [B, L] batch."""
        starts = self._start_table(int(input_ids.max()))[input_ids]
        c = np.cumsum(starts, axis=1)
        is_pad = input_ids == self._pad_id
        base = np.maximum.accumulate(np.where(is_pad, c, 0), axis=1)
        ww = c - base
        ww[:, -1] = 0
        return ww

    def _common(self, batch):
        input_ids, input_attention = _encode(self.tokenizer, [b["input"] for b in batch])
        output_ids, output_attention = _encode(self.tokenizer, [b["output"] for b in batch])
        ids = np.asarray(input_ids, dtype=np.int64)
        ww = self.whole_word_ids(ids)
        return (torch.from_numpy(ids), torch.tensor(input_attention), torch.from_numpy(ww), torch.tensor(output_ids),
                torch.tensor(output_attention))

    def __call__(self, batch):
        return self._common(batch)


class TestCollator(Collator):
    __test__ = False   # not a pytest class

    def __call__(self, batch):
        return self._common(batch) + (torch.tensor([b["user_idx"] for b in batch]),)
