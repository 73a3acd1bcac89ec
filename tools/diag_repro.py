"""Run-to-run reproducibility diagnostic (dev tool, GPU box).

mode `toy`:  the toy training of tests/test_gpu_runner.py::test_resume_on_device (ragged shapes, bf16, dropout on), RUNS straight
             runs in this process; after every optimizer step the CRC of every gradient tensor, of the parameter arena and of the
             first moments is recorded.  The report names the first step at which two runs differ and the tensors that differ there.
mode `c2`:   STEPS training steps on the benchmark shape (T5-small, B=64, L=128, T=8, dropout 0.1) on fresh models.
Every process also writes its record to gpurun_out/repro/<tag>.json so that fresh processes can be compared (mode `cmp`).
"""
import json
import os
import random
import sys
import tempfile
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ATOMIC = lambda n: n == "shared.weight" or "whole_word" in n or "relative_attention_bias" in n or "layer_norm" in n  # noqa: E731


def crc_views(flat_cpu, views):
    raw = flat_cpu.numpy().view("uint8")
    return {n: zlib.crc32(raw[4 * o:4 * (o + k)].tobytes()) for n, (o, k, _) in views.items()}


def record_step(rec, model, opt, keep_grads):
    g = model._grads.detach().cpu()
    rec["grad_crc"].append(crc_views(g, model._views))
    rec["param_crc"].append(zlib.crc32(model._flat.detach().cpu().numpy().tobytes()))
    rec["m_crc"].append(zlib.crc32(opt.m.detach().cpu().numpy().tobytes()))
    if keep_grads is not None:
        keep_grads.append(g)


def toy_run(hip, tmp, dropout, epochs=2):
    from torch.utils.data import ConcatDataset, DataLoader
    from openp5_amd import runner as R
    from openp5_amd.collator import Collator
    from openp5_amd.data import MultiTaskDataset
    from openp5_amd.sampler import SingleMultiDataTaskSampler
    from openp5_amd.tokenizer import build_offline_tokenizer
    from tests.test_host import SMALL_TOY, make_args
    from tests.test_runner_emu import VOCAB, tiny_model
    tok = build_offline_tokenizer(VOCAB)
    args = make_args(tmp, ["--epochs", str(epochs), "--test_before_train", "0", "--test_epoch", "0", "--batch_size", "8", "--sample_num", "1,1",
                           "--max_his", "3", "--lr", "3e-3"], toy=SMALL_TOY)
    args.model_path = os.path.join(tmp, "m.pt")
    random.seed(0)
    train = ConcatDataset([MultiTaskDataset(args, "Toy", "train")])
    loader = DataLoader(train, sampler=SingleMultiDataTaskSampler(train, args.batch_size, args.seed), batch_size=args.batch_size, collate_fn=Collator(tok))
    model = tiny_model(hip, len(tok), dropout=dropout, seed=9, dtype="bf16")
    model.set_dropout_seed(77, 0)
    runner = R.DistributedRunner(model, tok, loader, None, hip.device, args, 0)
    rec = {"grad_crc": [], "param_crc": [], "m_crc": [], "shapes": []}
    grads = []
    orig = R.training_step

    def hooked(model_, optimizer, batch, *a, **k):
        out = orig(model_, optimizer, batch, *a, **k)
        torch.cuda.synchronize()
        rec["shapes"].append([int(batch[0].shape[0]), int(batch[0].shape[1]), int(batch[3].shape[1])])
        record_step(rec, model_, optimizer, grads if len(grads) < 4 else None)
        return out

    R.training_step = hooked
    try:
        runner.train()
    finally:
        R.training_step = orig
    rec["final"] = model._flat.detach().cpu()
    rec["first_grads"] = grads
    rec["views"] = {n: [o, k] for n, (o, k, _) in model._views.items()}
    return rec


def pipe_run(hip, tmp, epochs=2):
    """the dataset gate's training (tests/test_gpu_dataset.py: cases.make_pipeline, T5-small dims, B=32, ragged L, random_initialization)"""
    from openp5_amd import runner as R
    from tests import cases
    runner, model, tok, args = cases.make_pipeline(hip, tmp, "bf16", flags=["--epochs", str(epochs), "--lr", "1e-3"])
    rec = {"grad_crc": [], "param_crc": [], "m_crc": [], "shapes": []}
    rec["init_crc"] = zlib.crc32(model._flat.detach().cpu().numpy().tobytes())
    grads = []
    orig = R.training_step

    def hooked(model_, optimizer, batch, *a, **k):
        out = orig(model_, optimizer, batch, *a, **k)
        torch.cuda.synchronize()
        rec["shapes"].append([int(batch[0].shape[0]), int(batch[0].shape[1]), int(batch[3].shape[1])])
        rec["in_crc"] = rec.get("in_crc", []) + [zlib.crc32(batch[0].cpu().numpy().tobytes()) ^ zlib.crc32(batch[3].cpu().numpy().tobytes())]
        record_step(rec, model_, optimizer, grads if len(grads) < 2 else None)
        return out

    R.training_step = hooked
    try:
        runner.train()
    finally:
        R.training_step = orig
    rec["final"] = model._flat.detach().cpu()
    rec["first_grads"] = grads
    rec["views"] = {n: [o, k] for n, (o, k, _) in model._views.items()}
    return rec


def c2_run(hip, steps, dropout=0.1, B=64, L=128, T=8):
    from oracle import t5_oracle as O
    from openp5_amd.optim import FusedAdamW
    from tests import cases
    ocfg = O.T5Cfg.named("t5-small", dropout=dropout)
    params = O.init_params(ocfg, 7)
    m = cases.build_model(hip, ocfg, params, "bf16", dropout)
    m.train()
    m.set_dropout_seed(5, 0)
    opt = FusedAdamW(m, lr=1e-3, max_grad_norm=1.0)
    rec = {"grad_crc": [], "param_crc": [], "m_crc": [], "shapes": []}
    grads = []
    for s in range(steps):
        a = [t.to(hip.device) for t in cases.synth_batch(ocfg, B, L, T, 3 + s)]
        m.loss_and_backward(*a)
        opt.step()
        m.zero_grad()
        torch.cuda.synchronize()
        rec["shapes"].append([B, L, T])
        record_step(rec, m, opt, grads if len(grads) < 2 else None)
    rec["final"] = m._flat.detach().cpu()
    rec["first_grads"] = grads
    rec["views"] = {n: [o, k] for n, (o, k, _) in m._views.items()}
    return rec


def compare(a, b, la, lb):
    if "init_crc" in a and "init_crc" in b:
        print(f"  {la} vs {lb}: initial parameters equal: {a['init_crc'] == b['init_crc']}; inputs equal: {a.get('in_crc') == b.get('in_crc')}")
    n = min(len(a["grad_crc"]), len(b["grad_crc"]))
    first = None
    for s in range(n):
        if a["grad_crc"][s] != b["grad_crc"][s]:
            first = s
            break
    fd = float((a["final"] - b["final"]).abs().max())
    if first is None:
        pd = [s for s in range(n) if a["param_crc"][s] != b["param_crc"][s]]
        print(f"  {la} vs {lb}: all {n} steps' gradients bit-identical; params differ at steps {pd[:5]}; final max |dparam| {fd:.3e}")
        return
    bad = [k for k in a["grad_crc"][first] if a["grad_crc"][first][k] != b["grad_crc"][first][k]]
    nat = [k for k in bad if not ATOMIC(k)]
    print(f"  {la} vs {lb}: first differing gradient at step {first} shape {a['shapes'][first]}: {len(bad)} tensors differ, {len(nat)} of them NOT sums of atomics; "
          f"params identical before it: {first == 0 or a['param_crc'][first - 1] == b['param_crc'][first - 1]}; final max |dparam| {fd:.3e}")
    print("     atomic-sum tensors: " + ", ".join(k.replace(".weight", "") for k in bad if ATOMIC(k))[:400])
    if nat:
        print("     NON-atomic tensors: " + ", ".join(k.replace(".weight", "") for k in nat)[:600])
    if first < len(a.get("first_grads", [])) and first < len(b.get("first_grads", [])):
        ga, gb = a["first_grads"][first], b["first_grads"][first]
        for k in bad[:40]:
            o, c = a["views"][k][:2]
            d = (ga[o:o + c] - gb[o:o + c]).abs()
            print(f"       {k}: max |d| {float(d.max()):.3e} over |g|max {float(ga[o:o + c].abs().max()):.3e}, {int((d > 0).sum())} of {c} elements")


def main():
    mode = sys.argv[1]
    if mode == "cmp":
        recs = []
        for p in sorted(sys.argv[2:]):
            recs.append((os.path.basename(p), torch.load(p)))
        for i in range(1, len(recs)):
            compare(recs[0][1], recs[i][1], recs[0][0], recs[i][0])
        return
    if os.environ.get("P5_DIAG_EMU"):        # (logic check of this tool on the host emulation)
        from tests.emu.emu_backend import emu_backend
        hip = emu_backend()
        torch.cuda.synchronize = lambda: None
    else:
        from openp5_amd._lib import hip_backend
        hip = hip_backend()
    tag = sys.argv[2]
    runs = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    dropout = float(sys.argv[4]) if len(sys.argv) > 4 else 0.1
    out = []
    for r in range(runs):
        if mode == "toy":
            with tempfile.TemporaryDirectory() as tmp:
                out.append(toy_run(hip, tmp, dropout))
        elif mode == "pipe":
            with tempfile.TemporaryDirectory() as tmp:
                out.append(pipe_run(hip, tmp))
                print(f"   run {r}: init crc {out[-1]['init_crc']}, {len(out[-1]['grad_crc'])} steps, input crcs equal to run 0: {out[-1]['in_crc'] == out[0]['in_crc']}")
        else:
            out.append(c2_run(hip, 3, dropout))
    print(f"[{mode} {tag}] dropout {dropout}: {len(out[0]['grad_crc'])} steps per run")
    for r in range(1, runs):
        compare(out[0], out[r], "run0", f"run{r}")
    os.makedirs("gpurun_out/repro", exist_ok=True)
    keep = {k: out[0][k] for k in ("grad_crc", "param_crc", "m_crc", "shapes", "final", "views", "first_grads", "init_crc", "in_crc") if k in out[0]}
    torch.save(keep, f"gpurun_out/repro/{mode}_{tag}.pt")


if __name__ == "__main__":
    main()
