"""-m gpu: dataset-level evaluation parity -- north_star: "generated item-ID sequences match the HF-CPU reference within fp32
logit tolerance (ranked Hit@k identical)".  A T5-small-dims model is trained for a few epochs on a synthetic dataset through the
real pipeline (datasets -> sampler -> collator -> runner, bf16 engine), then every test user of both tasks is evaluated three
ways with the SAME weights: bf16 engine, fp32 engine, fp32 CPU oracle (restated HF beam search + Python trie callbacks)."""
import pytest

from oracle import t5_oracle as O
from tests import cases

pytestmark = pytest.mark.gpu


def test_dataset_level_hit_ndcg_equal_oracle(hip, tmp_path):
    """the body, its tolerances and what each assertion means: tests/cases.py::dataset_gate (the host emulation runs the same body on a
    tiny model in the CPU suite, tests/test_runner_emu.py::test_dataset_gate_body_on_emulator)"""
    cases.dataset_gate(hip, str(tmp_path), lambda v: O.T5Cfg.named("t5-small", dropout=0.0, vocab_size=v), K=10, min_users=200,
                       flags=["--epochs", "6", "--lr", "1e-3"])
