"""GPU: cost of the M-tail row of tiles of the LLM products (M = 31 x 380 = 11780 = 46 x 256 + 4) in the 4-wave kernel:
the same product with M = 11776 (46 full rows of tiles) beside it, interleaved."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_llm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
for (N, K) in [(28672, 4096), (4096, 28672), (14336, 4096), (4096, 14336), (4096, 4096), (6144, 4160)]:
    a = torch.randn(11780, K, device=dev).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    c = torch.empty(11780, N, device=dev, dtype=torch.bfloat16)
    ops.gemm_set_config(12)
    res = {11780: [], 11776: []}
    for rnd in range(5):
        for M in (11780, 11776):
            ops.gemm_nt(a[:M], b, out=c[:M])
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5):
                ops.gemm_nt(a[:M], b, out=c[:M])
            e.record()
            torch.cuda.synchronize()
            res[M].append(s.elapsed_time(e) * 200)
    ops.gemm_set_config(0)
    t0, t1 = sorted(res[11780])[2], sorted(res[11776])[2]
    ref = a[11776:].float() @ b.float().T
    err = float((c[11776:].float() - ref).abs().max() / ref.abs().max())
    print(f"N={N:6d} K={K:6d}: M=11780 {t0:8.1f} us ({2.0 * 11780 * N * K / t0 / 1e6:7.1f} TF)   M=11776 {t1:8.1f} us   tail costs {100 * (t0 / t1 - 1):5.2f} %"
          f" (a full row of tiles would be {100 / 46:.2f} %)   tail rows rel err {err:.1e}")
