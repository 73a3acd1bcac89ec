"""`random_initialization` (/root/reference/src/src_t5/utils/initialization.py:15-35): re-draw N(0,1) the embedding rows of
every piece that occurs in the tokenisation of the integers 0..29999, so numeric item-ID pieces start untrained."""
import torch


@torch.no_grad()
def random_initialization(model, tokenizer, backbone="t5"):
    ids = set()
    for x in range(30000):
        ids.update(t for t in tokenizer.encode(str(x)) if t not in (1, 3))   # drop </s> and the bare word-start piece
    if not ids:
        return model
    index = torch.tensor(sorted(ids), dtype=torch.long, device=model.shared.weight.device)
    model.shared.weight.data[index] = torch.randn(len(index), model.shared.weight.shape[1], device=index.device)
    if hasattr(model, "mark_params_updated"):
        model.mark_params_updated()
    return model
