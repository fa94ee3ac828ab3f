"""GPU: the Whisper-large-v3 encoder GEMMs of the C3 batch (M = 31 x 1500) WITH their epilogues, per tile config."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_llm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M = 46500
cases = [("qkv  bias", 3840, 1280, dict(bias=True)), ("out  bias+res", 1280, 1280, dict(bias=True, res=True)),
         ("fc1  bias+gelu", 5120, 1280, dict(bias=True, act=ops.ACT_GELU)), ("fc2  bias+res", 1280, 5120, dict(bias=True, res=True))]
cfgs = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "1,6,7".split(","))]
for name, N, K, ep in cases:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    bias = torch.randn(N, device=dev) if ep.get("bias") else None
    res = torch.randn(M, N, device=dev).to(torch.bfloat16) if ep.get("res") else None
    best = {cfg: 1e9 for cfg in cfgs}
    for rnd in range(4):
        for cfg in cfgs:
            ops.gemm_set_config(cfg)
            ops.gemm_nt(a, b, out=c, bias=bias, residual=res, act=ep.get("act", ops.ACT_NONE))
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5):
                ops.gemm_nt(a, b, out=c, bias=bias, residual=res, act=ep.get("act", ops.ACT_NONE))
            e.record()
            torch.cuda.synchronize()
            best[cfg] = min(best[cfg], s.elapsed_time(e) / 5)
    ops.gemm_set_config(0)
    print(name, f"{M}x{N}x{K}", {f"cfg{cfg}": f"{best[cfg] * 1e3:.0f}us {2.0 * M * N * K / (best[cfg] * 1e-3) / 1e12:.0f}TF" for cfg in cfgs}, flush=True)
