import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_llm_amd import ops
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 6
M, N, K = 11780, 4096, 4096
dev = torch.device("cuda:0")
a = torch.randn(M, K, device=dev).to(torch.bfloat16); b = torch.randn(N, K, device=dev).to(torch.bfloat16)
c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
ops.gemm_set_config(cfg)
for _ in range(5):
    ops.gemm_nt(a, b, out=c)
torch.cuda.synchronize()
