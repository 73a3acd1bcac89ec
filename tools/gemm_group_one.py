"""Launches the step's dominant kernel alone a few times -- the eight weight gradients of two T5-small encoder layers over 8192 tokens as one
grouped launch (bench.py: time_wgrad_group) -- target for rocprofv3 --pmc (profiles/collect_pmc.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from openp5_amd._lib import hip_backend
be = hip_backend()
t, fl, by = bench.time_wgrad_group(be, 8192, 512, 2048, 512, iters=10)
print(f"{t * 1e6:.1f} us  {fl / t / 1e12:.0f} TFLOP/s")
