import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from slam_llm_amd import ops
dev = torch.device("cuda:0")
M, F = 11780, 14336
gu = torch.randn(M, 2 * F, device=dev).to(torch.bfloat16)
dh = torch.randn(M, F, device=dev).to(torch.bfloat16)
h = torch.empty(M, F, device=dev, dtype=torch.bfloat16)
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n
a = t(lambda: ops.swiglu_fwd(gu, out=h))
b = t(lambda: ops.swiglu_bwd(gu, dh))
print(f"swiglu_fwd {a:.1f} us {M*F*6/a/1e6:.2f} TB/s   swiglu_bwd {b:.1f} us {M*F*10/b/1e6:.2f} TB/s")
ref = torch.nn.functional.silu(gu[:, :F].float()) * gu[:, F:].float()
print("fwd max rel err", float(((h.float() - ref).abs() / (ref.abs() + 1e-3)).max()))
