"""GPU: per-shape GEMM rates INSIDE the C3 training step (HIP events), to compare with tools/gemm_bench.py (isolated)."""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SLAM_TIMER_SHAPES"] = "1"
import torch  # noqa: E402
from slam_llm_amd import ops  # noqa: E402

ops.TIMER_SHAPES = True
import bench  # noqa: E402

sys.argv = ["bench.py", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
import io, contextlib  # noqa: E402
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
rows = sorted(((k, v) for k, v in d["kernels"].items() if "[" in k), key=lambda kv: -kv[1]["ms_per_step"])
print(f"step {d['ms_per_step']:.1f} ms")
for k, v in rows[:24]:
    print(f"{k:70s} n={v['launches_per_step']:5.1f} ms/step={v['ms_per_step']:7.2f} TF={v['TFLOPs']:7.1f}")
