import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_llm_amd import ops
dev = torch.device("cuda:0")
B, T, H, D = 8, 1500, 20, 64
qkv = torch.randn(B * T, 3 * H * D, device=dev).to(torch.bfloat16)
vt = ops.head_rope_transpose(qkv, 2 * H * D, B, T, H, D)
o = torch.empty(B * T, H * D, device=dev, dtype=torch.bfloat16)
for _ in range(5):
    ops.attn_fwd(qkv[:, :H * D], qkv[:, H * D:2 * H * D], vt, B, T, H, H, D, False, D ** -0.5, want_lse=False, out=o)
torch.cuda.synchronize()
