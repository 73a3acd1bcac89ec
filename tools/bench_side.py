"""bench.py with the engine-internal side stream ENABLED (round 1/2 default; dev tool for the A/B against the single-stream default)."""
import os, runpy, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import openp5_amd.model as m
m.P5T5Native.use_side_stream = True
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
