"""GPU (tools): bench.py started with torch ALREADY imported, HIP_FORCE_DEV_KERNARG not exported: does the package's `os.environ.setdefault` still reach the HIP runtime?
python tools/kernarg_after_import.py --workload c4 --steps 16 --warmup 4 --no-cpu-baseline"""
import os
import runpy
import sys

os.environ.pop("HIP_FORCE_DEV_KERNARG", None)
import torch  # noqa: E402,F401  (loads libamdhip64; no HIP call yet)

sys.argv = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
