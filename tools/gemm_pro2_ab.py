"""GPU: A/B of the 4-wave GEMM's prologue -- one k-tile in flight before the first wait (slam_gemm_set_config 500) vs two (501,
shipped) -- on the C3 step's product shapes.  Interleaved, HIP events, median of 7 x 3 launches; outputs must be bit-identical."""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from slam_llm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [(11780, 6144, 4160), (11780, 4096, 4096), (11780, 28672, 4096), (11780, 4096, 14336), (11780, 14336, 4096), (4096, 128256, 4096)]


def timed(fn, n=3):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for M, N, K in SHAPES:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    outs, res = {}, {500: [], 501: []}
    try:
        ops.gemm_set_config(12)
        for mode in (500, 501):
            ops.call("slam_gemm_set_config", mode)
            c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            ops.gemm_nt(a, b, out=c)
            outs[mode] = c
        for _ in range(7):
            for mode in (500, 501):
                ops.call("slam_gemm_set_config", mode)
                res[mode].append(timed(lambda: ops.gemm_nt(a, b, out=outs[mode])))
    finally:
        ops.call("slam_gemm_set_config", 501)
        ops.gemm_set_config(0)
    m0, m1 = statistics.median(res[500]), statistics.median(res[501])
    print(json.dumps(dict(M=M, N=N, K=K, one_tile_us=round(m0, 1), two_tiles_us=round(m1, 1), speedup=round(m0 / m1, 4),
                          identical=bool(torch.equal(outs[500], outs[501])))), flush=True)
    del a, b, outs
