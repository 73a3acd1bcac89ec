"""TEST-ONLY: binds tests/emu/libp5emu.so (the kernel sources compiled for the host against hip_emu.h) behind the
same ctypes table as the product library, so the not-gpu suite can drive every kernel and the engine on CPU
tensors.  Never imported by openp5_amd."""
import ctypes
import os
import subprocess

import torch

from openp5_amd import _abi
from openp5_amd._lib import Backend

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libp5emu.so")
_BACKEND = None


def _stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    src = os.path.join(HERE, "..", "..", "openp5_amd", "csrc")
    files = [os.path.join(src, f) for f in os.listdir(src)] + [os.path.join(HERE, f) for f in ("hip_emu.h", "hip_emu.cpp")]
    files.append(os.path.join(HERE, "..", "..", "include", "p5hip.h"))
    return any(os.path.getmtime(f) > t for f in files)


def emu_backend():
    global _BACKEND
    if _BACKEND is None:
        if _stale():
            subprocess.check_call(["bash", os.path.join(HERE, "build_emu.sh")])
        lib = _abi.bind(ctypes.CDLL(SO))
        assert lib.p5_is_emulator() == 1
        _BACKEND = Backend(lib, torch.device("cpu"), True)
    return _BACKEND
