"""GPU: LoRA-side kernels at the C3 shape (M = 11780, K = 4096, r_total = 32): first hop, dA / dB grams + reduce, dropout."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_llm_amd import ops
dev = torch.device("cuda:0")
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n
M, K, R = 11780, 4096, 32
x = torch.randn(M, K + 64, device=dev).to(torch.bfloat16); A = torch.randn(R, K, device=dev).to(torch.bfloat16)
drop = (0.05, 1234, 1 << 40)
print(f"lora_a_fwd (dropout)   {t(lambda: ops.lora_a_fwd(x[:, :K], A, x[:, K:K + R], drop)):.1f} us")
print(f"lora_a_fwd (no drop)   {t(lambda: ops.lora_a_fwd(x[:, :K], A, x[:, K:K + R], None)):.1f} us")
du = torch.randn(M, R, device=dev).to(torch.bfloat16); gA = torch.zeros(R, K, device=dev)
print(f"dA gram+reduce (drop)  {t(lambda: ops.skinny_gram(du, x[:, :K], gA, K, 1, drop=drop)):.1f} us")
dy = torch.randn(M, 4096, device=dev).to(torch.bfloat16); u = torch.randn(M, 16, device=dev).to(torch.bfloat16); gB = torch.zeros(4096, 16, device=dev)
print(f"dB gram+reduce C=4096  {t(lambda: ops.skinny_gram(u, dy, gB, 1, 16)):.1f} us")
dyv = torch.randn(M, 1024, device=dev).to(torch.bfloat16); gBv = torch.zeros(1024, 16, device=dev)
print(f"dB gram+reduce C=1024  {t(lambda: ops.skinny_gram(u, dyv, gBv, 1, 16)):.1f} us")
h = torch.randn(M, K, device=dev).to(torch.bfloat16); acc = torch.zeros(M, K, device=dev).to(torch.bfloat16)
print(f"dropout accumulate     {t(lambda: ops.dropout(h, *drop, out=acc, accumulate=True)):.1f} us")
