"""not-gpu: the product C-ABI library builds for gfx950, loads, and exports every symbol include/p5hip.h declares
(no compute calls without a GPU); the product package refuses to run without a HIP device."""
import ctypes
import os
import re
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    path = os.path.join(ROOT, "openp5_amd", "libp5hip.so")
    if not os.path.exists(path):
        if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
            pytest.skip("hipcc not available")
        import __graft_entry__ as g
        g.build()
    return path


def test_header_symbols_exported(lib_path):
    from openp5_amd import _abi
    header = open(os.path.join(ROOT, "include", "p5hip.h")).read()
    declared = set(re.findall(r"\b(p5_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_abi.PROTOTYPES), declared ^ set(_abi.PROTOTYPES)
    lib = ctypes.CDLL(lib_path)
    for name in declared:
        assert hasattr(lib, name), name
    _abi.bind(lib)
    assert lib.p5_is_emulator() == 0 and lib.p5_abi_version() == 5


def test_engine_layout_without_gpu(lib_path):
    """p5_engine_create / p5_param_table are host-only: the arena layout must list the HF state-dict tensors."""
    from openp5_amd import _abi
    lib = _abi.bind(ctypes.CDLL(lib_path))
    cfg = _abi.P5Config(vocab_size=32100, d_model=512, d_kv=64, d_ff=2048, n_enc_layers=6, n_dec_layers=6, n_heads=8, rel_buckets=32,
                        rel_max_distance=128, whole_word_size=512, gated_gelu=0, dtype=1, eps=1e-6, dropout=0.1, pad_id=0, eos_id=1)
    eng = ctypes.c_void_p()
    assert lib.p5_engine_create(ctypes.byref(cfg), ctypes.byref(eng)) == 0
    names, total = [], 0
    name = ctypes.create_string_buffer(256)
    off, rows, cols = ctypes.c_int64(), ctypes.c_int(), ctypes.c_int()
    i = 0
    while lib.p5_param_table(eng, i, name, 256, ctypes.byref(off), ctypes.byref(rows), ctypes.byref(cols)) == 0:
        names.append(name.value.decode())
        total += rows.value * cols.value
        i += 1
    assert len(names) == 132 and total == 60_754_432            # t5-small at V=32100 + whole-word table
    assert "shared.weight" in names and "decoder.block.5.layer.1.EncDecAttention.k.weight" in names
    assert lib.p5_param_count(eng) >= total
    bad = _abi.P5Config(vocab_size=100, d_model=512, d_kv=32, d_ff=2048, n_enc_layers=1, n_dec_layers=1, n_heads=8, rel_buckets=32,
                        rel_max_distance=128, whole_word_size=512, gated_gelu=0, dtype=1, eps=1e-6, dropout=0.1, pad_id=0, eos_id=1)
    assert lib.p5_engine_create(ctypes.byref(bad), ctypes.byref(ctypes.c_void_p())) != 0
    assert b"d_kv" in lib.p5_last_error()
    lib.p5_engine_destroy(eng)


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from openp5_amd._lib import hip_backend
    from openp5_amd.model import P5ModelConfig, P5T5Native
    with pytest.raises(RuntimeError):
        hip_backend()
    with pytest.raises(RuntimeError):
        P5T5Native(P5ModelConfig(vocab_size=100))
