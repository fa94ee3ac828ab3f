"""GPU (tools): a few launches of slam_gemm_bf16_nt at the C3 step's dominant shapes under the AUTO kernel rule, for rocprofv3 --pmc passes.
python tools/pmc_gemm.py [n_launches]   (SLAM_HIP_LIB selects another build of the library for A/B arms)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from slam_llm_amd import ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda:0")
for (M, N, K) in ((11780, 28672, 4096), (11780, 4096, 4096), (11780, 4096, 14336), (46500, 5120, 1280)):
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    a = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).to(torch.bfloat16)
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(n):
        ops.gemm_nt(a, b, out=c)
    torch.cuda.synchronize()
