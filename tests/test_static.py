"""not-gpu: source-level guards.

The training step is bit-reproducible because no gradient is a sum of fp32 atomics any more.  A float `atomicAdd` that
creeps back into a training kernel would not fail any tolerance-based parity test -- it would only make two runs differ in their last
bits, which AdamW then amplifies -- and the GPU reproducibility tests would catch it a round later.  This test catches it at once: every
float atomic in the kernel sources must be on the short list of known, non-default or generation-only sites."""
import os
import re

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "openp5_amd", "csrc")

# file -> substrings identifying the lines that MAY hold an fp32 / LDS float atomic, with the reason
ALLOWED = {
    "p5_elem.h": ["atomicAdd(dE + ", "atomicAdd(dWW + ",      # p5_embed_bwd_kernel: the atomic scatter of rounds 1-3, only with p5_set_option("embed_det", 0)
                  "atomicAdd(dw + j, v)"],                     # p5_rmsnorm_bwd_kernel without a partial buffer: the stand-alone op entry's legacy mode
    "p5_gemm.h": ["atomicAdd(essq + row, ss)", "atomicAdd(crow + col, v)", "atomicAdd(g.ssq_out + row, w * w)", "atomicAdd(((float*)g.C) + ci, v)"],
    "p5_gemm4.h": ["atomicAdd(cp + r, v[r])", "atomicAdd((float*)g.C + ci + r, v[r])", "atomicAdd(cp + e, v[e])", "atomicAdd(g.ssq_out + row, ss)"],
    "p5_gemm5.h": ["atomicAdd(cp + r, v[r])", "atomicAdd((float*)g.C + ci + r, v[r])", "atomicAdd(cp + e, v[e])", "atomicAdd(g.ssq_out + row, ss)"],
    # (GEMM epilogues: P5_EPI_ATOMIC is issued by the engine with ONE split only -- each element receives a single add per backward --
    #  or with c_split_stride > 0, which stores; the scalar ssq form is the first-generation decode step's, generation only)
    "p5_decode2.h": ["atomicAdd((float*)g.C + ci, v)"],        # decode step of rounds 2-4 (K-split workgroups adding into the residual stream): only with p5_set_option("dec_atomic", 1)
}


def test_no_new_float_atomics_in_kernel_sources():
    found = {}
    for fn in sorted(os.listdir(CSRC)):
        if not fn.endswith((".h", ".hip")):
            continue
        for i, line in enumerate(open(os.path.join(CSRC, fn)), 1):
            code = line.split("//")[0]
            if "atomicAdd(" not in code:
                continue
            if re.search(r"atomicAdd\(&?(s_nothit|st\.flags|hist|s_sel|pl\.hdr)", code):       # integer counters of the beam search / verification plan
                continue
            if not any(tok in code for tok in ALLOWED.get(fn, [])):
                found.setdefault(fn, []).append((i, code.strip()))
    assert not found, f"float atomics outside the allowed sites (DESIGN.md 3.5): {found}"


def test_decode_step_updates_the_residual_stream_with_one_writer_per_element():
    src = open(os.path.join(CSRC, "p5_lib.hip")).read()
    assert "g_opt_dec_atomic ? P5_SK_ATOMIC : P5_SK_RESID" in src and 'P5_DEC_ATOMIC") ? atoi(getenv("P5_DEC_ATOMIC")) : 0' in src
    assert "P5_SK_ATOMIC, 1.f, 0.f, done" not in src         # no projection of the decode step asks for the atomic epilogue directly


def test_engine_issues_atomic_gemms_with_one_split():
    src = open(os.path.join(CSRC, "p5_lib.hip")).read()
    # the only P5_EPI_ATOMIC problems the engine builds: the ungrouped weight gradient (one split unless P5_WGRAD_SPLIT_ATOMIC) and the tied
    # head's input gradient (c_split_stride > 0: partial products are stored and summed in order)
    sites = [m.start() for m in re.finditer(r"g\.epi = P5_EPI_ATOMIC", src)]
    assert len(sites) == 2, sites
    assert "g.splitk = g_opt_wgrad_split_atomic ? 0 : 1" in src
    assert "g.c_split_stride = (long long)Md * d" in src
