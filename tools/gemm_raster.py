"""GPU: raster group height (M-tiles per group) sweep of the pipelined 256x256 GEMM on the big-N shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_llm_amd import ops
from slam_llm_amd.lib import call
dev = torch.device("cuda:0")
for (M, N, K) in [(11780, 28672, 4096), (4096, 128256, 4096), (11780, 4096, 4096), (11780, 4096, 14336), (46500, 5120, 1280)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    out = {}
    for rnd in range(3):
        for gm in (2, 4, 8, 16, 32, 47):
            call("slam_gemm_set_group_m", gm)
            ops.gemm_nt(a, b, out=c); torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5): ops.gemm_nt(a, b, out=c)
            e.record(); torch.cuda.synchronize()
            tf = 2.0 * M * N * K / (s.elapsed_time(e) / 5 * 1e-3) / 1e12
            out[gm] = max(out.get(gm, 0), round(tf))
    call("slam_gemm_set_group_m", 8)
    print((M, N, K), out, flush=True)
