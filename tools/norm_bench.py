"""GPU: RMSNorm fwd/bwd at the LLM shape and LayerNorm at the Whisper shape of the C3 batch."""
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
M, d = 11780, 4096
x = torch.randn(M, d, device=dev).to(torch.bfloat16); dy = torch.randn_like(x); dres = torch.randn_like(x)
w = torch.randn(d, device=dev); y = torch.empty_like(x); dx = torch.empty_like(x)
_, rstd = ops.rmsnorm_fwd(x, w, 1e-5, out=y)
us = t(lambda: ops.rmsnorm_fwd(x, w, 1e-5, out=y, rstd=rstd)); print(f"rmsnorm_fwd {us:.1f} us {2 * M * d * 2 / us / 1e6:.2f} TB/s")
us = t(lambda: ops.rmsnorm_bwd(x, rstd, w, dy, dres=dres, out=dx)); print(f"rmsnorm_bwd {us:.1f} us {4 * M * d * 2 / us / 1e6:.2f} TB/s")
M2, d2 = 46500, 1280
x2 = torch.randn(M2, d2, device=dev).to(torch.bfloat16); w2 = torch.randn(d2, device=dev); b2 = torch.randn(d2, device=dev)
us = t(lambda: ops.layernorm(x2, w2, b2, 1e-5)); print(f"layernorm {us:.1f} us {2 * M2 * d2 * 2 / us / 1e6:.2f} TB/s")
F_ = 14336
gu = torch.randn(M, 2 * F_, device=dev).to(torch.bfloat16); hh = torch.empty(M, F_, device=dev, dtype=torch.bfloat16); dgu = torch.empty_like(gu)
us = t(lambda: ops.swiglu_fwd(gu, out=hh)); print(f"swiglu_fwd {us:.1f} us {3 * M * F_ * 2 / us / 1e6:.2f} TB/s")
us = t(lambda: ops.swiglu_bwd(gu, hh, out=dgu)); print(f"swiglu_bwd {us:.1f} us {5 * M * F_ * 2 / us / 1e6:.2f} TB/s")
