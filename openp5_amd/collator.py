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


class WordCacheEncoder:
    """Tokenises prompts word by word with a per-word cache.  SentencePiece/T5 never merges across whitespace (every
    whitespace-separated word is segmented on its own, prefixed by U+2581), so concat(pieces(word)) + </s> equals the
    tokenizer's own output -- but OpenP5 prompts are built from a few hundred template words plus item / user ids that
    repeat all the time, so after warm-up a 64-prompt batch costs ~1 ms instead of ~8 ms of Unigram Viterbi (which would
    make the input pipeline slower than the MI355X training step).  `ok` is False when a
    self-check against the wrapped tokenizer fails (then the collator falls back to the tokenizer itself)."""

    PROBE = "Considering ML1M user_12 has interacted with ML1M items item_1001 , item_27 . What is next ?  <x> a,b"

    def __init__(self, tokenizer, max_length=512):
        self.tok, self.max_length = tokenizer, max_length
        self.eos = tokenizer.eos_token_id
        self.pad = getattr(tokenizer, "pad_token_id", 0) or 0
        self.cache = {}
        try:
            self.ok = self.eos is not None and self.encode_one(self.PROBE) == tokenizer(self.PROBE, truncation=True, max_length=max_length)["input_ids"]
        except Exception:
            self.ok = False

    def _word(self, w):
        ids = self.cache.get(w)
        if ids is None:
            ids = self.tok.encode(w, add_special_tokens=False)
            self.cache[w] = ids
        return ids

    def encode_one(self, text):
        out = []
        for w in text.split():
            out.extend(self._word(w))
        return out[: self.max_length - 1] + [self.eos]

    def __call__(self, texts):
        rows = [self.encode_one(t) for t in texts]
        L = max(len(r) for r in rows)
        ids = np.full((len(rows), L), self.pad, dtype=np.int64)
        mask = np.zeros((len(rows), L), dtype=np.int64)
        for i, r in enumerate(rows):
            ids[i, : len(r)] = r
            mask[i, : len(r)] = 1
        return ids, mask


class Collator:
    def __init__(self, tokenizer, fast=True, solo_rows=False):
        """solo_rows: compute whole-word ids as if every row were collated alone.  The reference zeroes only the last
        COLUMN of the padded batch, so its inputs depend on batch composition; the per-user filtered protocol is defined at
        batch size 1 (DistributedRunner.py:271-337), and this flag lets it be batched with bit-identical inputs."""
        self.tokenizer = tokenizer
        self.solo_rows = bool(solo_rows)
        self._starts = None
        self._pad_id = getattr(tokenizer, "pad_token_id", 0) or 0
        self._fast = WordCacheEncoder(tokenizer) if fast else None
        if self._fast is not None and not self._fast.ok:
            self._fast = None

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
        if self.solo_rows:     # what each row would get in a batch of its own: its last real token (</s>) is "the last column"
            n = (~is_pad).sum(axis=1)
            ww[np.arange(len(ww)), np.maximum(n, 1) - 1] = 0
        return ww

    def _common(self, batch):
        enc = self._fast if self._fast is not None else (lambda texts: tuple(np.asarray(x, dtype=np.int64) for x in _encode(self.tokenizer, texts)))
        ids, input_attention = enc([b["input"] for b in batch])
        output_ids, output_attention = enc([b["output"] for b in batch])
        ww = self.whole_word_ids(ids)
        return (torch.from_numpy(ids), torch.from_numpy(input_attention), torch.from_numpy(ww), torch.from_numpy(output_ids),
                torch.from_numpy(output_attention))

    def __call__(self, batch):
        return self._common(batch)


class TestCollator(Collator):
    __test__ = False   # not a pytest class

    def __call__(self, batch):
        return self._common(batch) + (torch.tensor([b["user_idx"] for b in batch]),)
