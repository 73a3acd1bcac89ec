"""Offline stand-in for `AutoTokenizer.from_pretrained("t5-small")` (main.py:171): there is no network and no
`spiece.model` on disk, so a Unigram T5Tokenizer is built from a synthetic piece list that keeps what the path depends
on -- pad=0, </s>=1, <unk>=2, 32,100 entries (so V, resize_token_embeddings and the lm_head shape match the real
vocabulary), word-start pieces, `item`, `_`, `user`, dataset names and 1-3 digit number pieces so that
"<dataset> item_<id>" splits into a handful of pieces like the real vocabulary (SURVEY.md 8(c) "Tokenizer stand-in")."""
import os

VOCAB_SIZE = 32100


def synthetic_pieces(vocab_size=VOCAB_SIZE, datasets=("ML1M", "ML100K", "Beauty", "Yelp", "Toy")):
    pieces = [("<pad>", 0.0), ("</s>", 0.0), ("<unk>", 0.0), ("▁", -2.0)]
    words = ["item", "user", "items", "What", "would", "be", "likely", "to", "purchase", "next", "after", "buying", "Considering", "has",
             "interacted", "with", "is", "the", "recommendation", "for", "?", ".", ",", ":", "Here", "history", "of", "I", "wonder",
             "what", "recommended", "predict", "possible", "bought", "by", "find", "list", "other", "does", "need", "Can", "you", "help",
             "me", "decide", "According", "purchased", "recommend", "another", "should", "we", "looking", "some", "Do", "have", "any",
             "recommendations", "suggested", "choose", "an", "and", "a", "the"] + list(datasets)
    seen = set(p for p, _ in pieces)
    for w in words:
        for form in ("▁" + w, w):
            if form not in seen:
                pieces.append((form, -4.0))
                seen.add(form)
    pieces.append(("_", -3.0))
    for n in range(1000):                       # 1-3 digit number pieces (4-digit item ids split in two, as with the real vocabulary)
        for form in (str(n), "▁" + str(n)):
            if form not in seen and len(pieces) < vocab_size - 64:
                pieces.append((form, -6.0 - 0.0001 * n))
                seen.add(form)
    for ch in "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ<>":
        if ch not in seen:
            pieces.append((ch, -9.0))
            seen.add(ch)
    i = 0
    while len(pieces) < vocab_size:             # filler so that len(tokenizer) == vocab_size
        form = f"▁zz{i}"
        if form not in seen:
            pieces.append((form, -20.0))
        i += 1
    return pieces[:vocab_size]


def build_offline_tokenizer(vocab_size=VOCAB_SIZE, **kw):
    from transformers import T5Tokenizer
    tok = T5Tokenizer(vocab=synthetic_pieces(vocab_size, **kw), extra_ids=0)
    if not hasattr(tok, "batch_encode_plus"):
        tok.batch_encode_plus = tok.__call__
    return tok


def load_tokenizer(backbone="t5-small"):
    """A local HF tokenizer directory if `backbone` points to one, otherwise the synthetic offline tokenizer."""
    if os.path.isdir(str(backbone)):
        from transformers import AutoTokenizer
        return AutoTokenizer.from_pretrained(backbone)
    return build_offline_tokenizer()
