"""GPU, tools only: where is the ceiling of the dominant GEMM?  (the "measured proof of ceiling" branch.)

For the eight large-K products of the C3 step (M = 11 780 rows; forward and dX of qkv / o / gate|up / down) and the encoder products:
  * this library's kernel under the auto rule on random operands (what the step runs),
  * the same launch on ZERO operands (no toggling in the multipliers: the clock the part sustains when power does not bind -- the gap
    between the two lines is the power limit, not the instruction stream),
  * hipBLASLt behind torch.matmul on the same operands (external bar; NEVER linked into the product, never imported by slam_llm_amd).
Interleaved rounds, best of 5 x 8 launches each.  Writes a markdown table to stdout.
    python tools/gemm_ceiling.py > profiles/r06_gemm_ceiling.md"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_llm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M = 31 * 380
SHAPES = [("qkv fwd", M, 6144, 4160), ("o fwd / dX", M, 4096, 4096), ("gate|up fwd", M, 28672, 4096), ("down fwd", M, 4096, 14336),
          ("qkv dX", M, 4096, 6144), ("gate|up dX", M, 4096, 28672), ("down dX", M, 14336, 4096), ("lm_head (label rows)", 1984, 128256, 4096),
          ("enc qkv", 46500, 3840, 1280), ("enc out", 46500, 1280, 1280), ("enc fc1", 46500, 5120, 1280), ("enc fc2", 46500, 1280, 5120)]


def timed(fn, reps=8):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


print("# GEMM ceiling at HEAD (tools/gemm_ceiling.py; 1 MI355X; TFLOP/s, best of 5 interleaved rounds x 8 launches)\n")
print("| product | M x N x K | kernel (auto rule) | this library, random | this library, zeros | hipBLASLt, random | hipBLASLt, zeros | ours / hipBLASLt (random) | random / zeros (ours) |")
print("|---|---|---|---|---|---|---|---|---|")
ratios = []
for name, m, n, k in SHAPES:
    g = torch.Generator(device=dev).manual_seed(1)
    a = torch.randn(m, k, generator=g, device=dev).to(torch.bfloat16)
    b = (torch.randn(n, k, generator=g, device=dev) * k ** -0.5).to(torch.bfloat16)
    az, bz = torch.zeros_like(a), torch.zeros_like(b)
    c = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    forms = {"ours": lambda: ops.gemm_nt(a, b, out=c), "ours0": lambda: ops.gemm_nt(az, bz, out=c),
             "lt": lambda: torch.matmul(a, b.T, out=c), "lt0": lambda: torch.matmul(az, bz.T, out=c)}
    best = {key: 1e9 for key in forms}
    for key, fn in forms.items():
        fn()
    for rnd in range(5):
        for key, fn in forms.items():
            best[key] = min(best[key], timed(fn))
    tf = {key: 2.0 * m * n * k / (v * 1e-3) / 1e12 for key, v in best.items()}
    ratios.append(tf["ours"] / tf["lt"])
    print(f"| {name} | {m} x {n} x {k} | {ops.gemm_kernel_name(m, n, k)} | {tf['ours']:.0f} | {tf['ours0']:.0f} | {tf['lt']:.0f} | {tf['lt0']:.0f} | "
          f"{tf['ours'] / tf['lt']:.3f} | {tf['ours'] / tf['ours0']:.3f} |")
    del a, b, az, bz, c
print(f"\nours / hipBLASLt on random operands: min {min(ratios):.3f}, geometric mean {torch.tensor(ratios).log().mean().exp().item():.3f}, max {max(ratios):.3f}.")
print("Peak for the roofline: 2 500 TFLOP/s dense bf16.  Zero operands remove the multipliers' toggling: what is left of the gap to the peak on that "
      "line is instruction stream + tile quantisation; the gap BETWEEN the random and the zero line is the power limit.")
