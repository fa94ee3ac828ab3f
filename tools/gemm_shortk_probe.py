"""GPU: the Whisper-encoder GEMM shapes (K = 1280) per tile config: rate, and for the non-persistent kernels the prologue / k-loop /
epilogue shader cycles of workgroup 0's tile plus the effective clock (the persistent kernel, cfg 7, writes no stamps)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from slam_llm_amd import ops
ops.call("slam_gemm_set_config", 401)   # workgroup 0 of cfg 6 / 12 launches stamps its phases (PROBE instantiation; off in production)
dev = torch.device("cuda:0")
for (M, N, K, ep) in ((46500, 3840, 1280, {}), (46500, 3840, 1280, dict(bias=True)), (46500, 5120, 1280, dict(bias=True, act=ops.ACT_GELU)), (46500, 1280, 1280, dict(bias=True, res=True))):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    bias = torch.randn(N, device=dev) if ep.get("bias") else None
    res = torch.randn(M, N, device=dev).to(torch.bfloat16) if ep.get("res") else None
    for CFG in (7, 6, 12):
        ops.gemm_set_config(CFG)
        for _ in range(3):
            ops.gemm_nt(a, b, out=c, bias=bias, residual=res, act=ep.get("act", ops.ACT_NONE))
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            ops.gemm_nt(a, b, out=c, bias=bias, residual=res, act=ep.get("act", ops.ACT_NONE))
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) * 200
        clk = (ctypes.c_ulonglong * 6)()
        ops.call("slam_gemm_debug_clock", ctypes.cast(clk, ctypes.c_void_p))
        ghz = (clk[2] - clk[0]) / max(1, (clk[3] - clk[1])) / 10.0
        tiles = ((M + 255) // 256) * ((N + 255) // 256)
        if CFG == 7:
            print(f"{M}x{N}x{K} {sorted(ep)} cfg {CFG}: {us:7.1f} us {2.0*M*N*K/us/1e6:7.1f} TF | (persistent: no stamps)")
            continue
        print(f"{M}x{N}x{K} {sorted(ep)} cfg {CFG}: {us:7.1f} us {2.0*M*N*K/us/1e6:7.1f} TF | WG0: prologue {clk[4]-clk[0]:6d} k-loop {clk[2]-clk[4]:7d} ({(clk[2]-clk[4])/(K//64):.0f}/k-tile) epilogue {clk[5]-clk[2]:6d} | {ghz:.2f} GHz | {tiles} tiles = {tiles/256:.2f} rounds -> {us*1e-6*ghz*1e9/ -(-tiles//256):.0f} cycles per round")
    ops.gemm_set_config(0)
