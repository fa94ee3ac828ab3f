"""GPU (tools): raster group height (M-tiles per XCD-local group) sweep on the Whisper-large encoder products of the C3 step (K = 1280 / 5120,
the persistent and the 8-wave pipelined kernels), with the fused epilogues the step uses.  python tools/gemm_enc_raster.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_llm_amd import ops  # noqa: E402
from slam_llm_amd.lib import call  # noqa: E402

dev = torch.device("cuda:0")
M = 31 * 1500
SHAPES = [("qkv", M, 3840, 1280, True, False, False), ("out", M, 1280, 1280, True, False, True), ("fc1", M, 5120, 1280, True, True, False),
          ("fc2", M, 1280, 5120, True, False, True)]
ML = 31 * 380
if len(sys.argv) > 1 and sys.argv[1] == "llm":      # the large-K products of the LLM (no epilogue arguments: the raster does not depend on them)
    SHAPES = [("qkv fwd", ML, 6144, 4160, False, False, False), ("o", ML, 4096, 4096, False, False, True), ("gate|up fwd", ML, 28672, 4096, False, False, False),
              ("down fwd", ML, 4096, 14336, False, False, True), ("qkv dX", ML, 4096, 6144, False, False, False), ("gate|up dX", ML, 4096, 28672, False, False, False),
              ("down dX", ML, 14336, 4096, False, False, False), ("lm_head", 1984, 128256, 4096, False, False, False), ("lm_head dX", 1984, 4096, 128256, False, False, False)]
GMS = (2, 4, 8, 12, 16, 24, 32, 64)
for name, m, n, k, bias, gelu, res in SHAPES:
    g = torch.Generator(device=dev).manual_seed(1)
    a = torch.randn(m, k, generator=g, device=dev).to(torch.bfloat16)
    b = (torch.randn(n, k, generator=g, device=dev) * k ** -0.5).to(torch.bfloat16)
    bias_t = torch.randn(n, generator=g, device=dev) if bias else None
    res_t = torch.randn(m, n, generator=g, device=dev).to(torch.bfloat16) if res else None
    c = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    best = {gm: 1e9 for gm in GMS}
    for rnd in range(4):
        for gm in GMS:
            call("slam_gemm_set_group_m", gm)
            ops.gemm_nt(a, b, out=c, bias=bias_t, act=ops.ACT_GELU if gelu else ops.ACT_NONE, residual=res_t)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(8):
                ops.gemm_nt(a, b, out=c, bias=bias_t, act=ops.ACT_GELU if gelu else ops.ACT_NONE, residual=res_t)
            e.record()
            torch.cuda.synchronize()
            best[gm] = min(best[gm], s.elapsed_time(e) / 8 * 1e3)
    call("slam_gemm_set_group_m", 0)      # back to the per-shape rule
    print(json.dumps(dict(product=name, shape=[m, n, k], kernel=ops.gemm_kernel_name(m, n, k), us={gm: round(v, 1) for gm, v in best.items()},
                          TF={gm: round(2.0 * m * n * k / v / 1e6) for gm, v in best.items()})), flush=True)
