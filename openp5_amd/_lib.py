"""Loader for the HIP extension (openp5_amd/libp5hip.so, built by __graft_entry__.build()).

There is deliberately NO CPU fallback: if the shared library is missing, or it is not the gfx950 build,
importing the compute path raises.  (tests/emu builds a host emulation of the same kernel sources for the
not-gpu test-suite; it is injected explicitly by those tests and is never discovered from here.)
"""
import ctypes
import os

from . import _abi

_LIB = None
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libp5hip.so")


class Backend:
    """A bound C-ABI library plus the torch device its pointers live on."""

    def __init__(self, cdll, device, is_emulator):
        self.lib = cdll
        self.device = device
        self.is_emulator = is_emulator

    def stream_ptr(self):
        if self.is_emulator:
            return None
        import torch
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def check(self, rc, what=""):
        _abi.check(self.lib, rc, what)


def load_hip_library():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
                "openp5_amd has no CPU fallback.")
        # torch first: its wheel bundles its own libamdhip64; a process that loads this library (linked against the system HIP
        # runtime) BEFORE torch ends up with two HIP runtimes, and every launch from here then fails with "no ROCm-capable
        # device is detected" because device memory and streams belong to the other one
        import torch  # noqa: F401
        lib = _abi.bind(ctypes.CDLL(LIB_PATH))
        if lib.p5_is_emulator():
            raise ImportError("libp5hip.so is an emulator build; refusing to use it as the product library")
        _LIB = lib
    return _LIB


def hip_backend(device=None):
    import torch
    lib = load_hip_library()
    if not torch.cuda.is_available():
        raise RuntimeError("openp5_amd needs an MI355X (HIP device); torch.cuda.is_available() is False and there is no CPU fallback")
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return Backend(lib, torch.device(device), False)
