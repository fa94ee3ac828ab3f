"""GPU: prologue / k-loop / epilogue shader cycles of workgroup 0 of the pipelined GEMM (cfg 6) as the number of concurrently
running tiles grows -- is the epilogue's store tail a per-CU cost or a chip-wide burst?"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_llm_amd import ops
ops.call("slam_gemm_set_config", 401)   # workgroup 0 of cfg 6 / 12 launches stamps its phases (PROBE instantiation; off in production)
dev = torch.device("cuda:0")
K = 4096
CFG = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for tiles_m, tiles_n in ((1, 1), (4, 8), (16, 16), (46, 16)):
    M, N = tiles_m * 256, tiles_n * 256
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ops.gemm_set_config(CFG)
    for _ in range(3):
        ops.gemm_nt(a, b, out=c)
    torch.cuda.synchronize()
    clk = (ctypes.c_ulonglong * 6)()
    ops.call("slam_gemm_debug_clock", ctypes.cast(clk, ctypes.c_void_p))
    ops.gemm_set_config(0)
    print(f"cfg {CFG} {tiles_m * tiles_n:4d} tiles: prologue {clk[4] - clk[0]:6d}  k-loop {clk[2] - clk[4]:7d} ({(clk[2] - clk[4]) / (K // 64):.0f}/k-tile)  epilogue {clk[5] - clk[2]:6d} cycles")
