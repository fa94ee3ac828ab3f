"""GPU: products with a residual epilogue (Whisper out-proj / fc2, Llama o_proj / down_proj), auto config."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_llm_amd import ops
dev = torch.device("cuda:0")
for (M, N, K, bias) in ((46500, 1280, 1280, True), (46500, 1280, 5120, True), (11780, 4096, 4096, False), (11780, 4096, 14336, False)):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    res = torch.randn(M, N, device=dev).to(torch.bfloat16)
    bi = torch.randn(N, device=dev) if bias else None
    f = lambda: ops.gemm_nt(a, b, out=c, bias=bi, residual=res)
    for _ in range(3): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): f()
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 200)
    us = sorted(ts)[1]
    ref = (a[:64].float() @ b.float().T) + (bi if bias else 0) + res[:64].float()
    err = float((c[:64].float() - ref).abs().max() / ref.abs().max())
    print(f"{M}x{N}x{K} bias={bias} +res: {us:8.1f} us {2.0*M*N*K/us/1e6:7.1f} TF  ({ops.gemm_kernel_name(M, N, K)})  rel err {err:.1e}")
