"""Per-kernel timing on the GPU box (GEMM tile configs, attention, bandwidth kernels). Writes gpurun_out/perf_ops.json."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_llm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
res = []


def timeit(fn, iters=10, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def gemm_cases():
    shapes = [(11780, 6144, 4160), (11780, 4096, 4096), (11780, 28672, 4096), (11780, 4096, 14336),
              (46500, 3840, 1280), (46500, 5120, 1280), (46500, 1280, 5120), (4096, 128256, 4096),
              (9300, 2048, 6400), (4096, 4096, 4096), (8192, 8192, 8192)]
    for (M, N, K) in shapes:
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        b = torch.randn(N, K, device=dev).to(torch.bfloat16)
        c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        for cfg in (1, 2, 4):
            ops.gemm_set_config(cfg)
            try:
                t = timeit(lambda: ops.gemm_nt(a, b, out=c), iters=5)
                tf = 2.0 * M * N * K / t / 1e12
                res.append({"op": "gemm", "M": M, "N": N, "K": K, "cfg": cfg, "ms": t * 1e3, "TF": tf})
                print(res[-1], flush=True)
            except Exception as ex:  # noqa: BLE001
                print("gemm failed", M, N, K, cfg, ex, flush=True)
        ops.gemm_set_config(0)
        del a, b, c


def attn_cases():
    for (B, T, Hq, Hkv, D, causal, bwd) in [(8, 1500, 20, 20, 64, False, False), (31, 380, 32, 8, 128, True, True)]:
        ld = (Hq + 2 * Hkv) * D
        qkv = torch.randn(B * T, ld, device=dev).to(torch.bfloat16)
        qt = ops.head_rope_transpose(qkv, 0, B, T, Hq, D)
        kt = ops.head_rope_transpose(qkv, Hq * D, B, T, Hkv, D)
        vt = ops.head_rope_transpose(qkv, (Hq + Hkv) * D, B, T, Hkv, D)
        q2, k2, v2 = qkv[:, :Hq * D], qkv[:, Hq * D:(Hq + Hkv) * D], qkv[:, (Hq + Hkv) * D:]
        o = torch.empty(B * T, Hq * D, device=dev, dtype=torch.bfloat16)
        scale = D ** -0.5
        t = timeit(lambda: ops.attn_fwd(q2, k2, vt, B, T, Hq, Hkv, D, causal, scale, out=o))
        fl = 4.0 * B * Hq * T * T * D * (0.5 if causal else 1.0)
        res.append({"op": "attn_fwd", "B": B, "T": T, "Hq": Hq, "D": D, "causal": causal, "ms": t * 1e3, "TF": fl / t / 1e12})
        print(res[-1], flush=True)
        if bwd:
            _, lse = ops.attn_fwd(q2, k2, vt, B, T, Hq, Hkv, D, causal, scale, out=o)
            do = torch.randn_like(o)
            dot = ops.head_rope_transpose(do, 0, B, T, Hq, D)
            dqkv = torch.empty_like(qkv)
            dq2, dk2, dv2 = dqkv[:, :Hq * D], dqkv[:, Hq * D:(Hq + Hkv) * D], dqkv[:, (Hq + Hkv) * D:]
            t = timeit(lambda: ops.attn_bwd(q2, k2, v2, o, do, lse, dq2, dk2, dv2, B, T, Hq, Hkv, D, causal, scale))
            res.append({"op": "attn_bwd", "B": B, "T": T, "ms": t * 1e3, "TF": 2.5 * fl / t / 1e12})
            print(res[-1], flush=True)


def bw_cases():
    M, d = 11780, 4096
    x = torch.randn(M, d, device=dev).to(torch.bfloat16)
    w = torch.ones(d, device=dev)
    y = torch.empty_like(x)
    rstd = torch.empty(M, device=dev)
    t = timeit(lambda: ops.rmsnorm_fwd(x, w, 1e-5, out=y, rstd=rstd))
    res.append({"op": "rmsnorm_fwd", "ms": t * 1e3, "GBs": 2 * M * d * 2 / t / 1e9}); print(res[-1], flush=True)
    gu = torch.randn(M, 28672, device=dev).to(torch.bfloat16)
    h = torch.empty(M, 14336, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: ops.swiglu_fwd(gu, out=h))
    res.append({"op": "swiglu_fwd", "ms": t * 1e3, "GBs": 3 * M * 14336 * 2 / t / 1e9}); print(res[-1], flush=True)
    audio = torch.randn(31, 480000, device=dev) * 0.1
    t = timeit(lambda: ops.logmel(audio, 128), iters=3)
    res.append({"op": "logmel", "clips": 31, "ms": t * 1e3, "GBs": 31 * 4 * (480000 + 128 * 3000) / t / 1e9}); print(res[-1], flush=True)
    xe = torch.randn(46500, 1280, device=dev).to(torch.bfloat16)
    wl, bl = torch.ones(1280, device=dev), torch.zeros(1280, device=dev)
    ye = torch.empty_like(xe)
    t = timeit(lambda: ops.layernorm(xe, wl, bl, out=ye))
    res.append({"op": "layernorm", "ms": t * 1e3, "GBs": 2 * 46500 * 1280 * 2 / t / 1e9}); print(res[-1], flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["gemm", "attn", "bw"]
    for w in which:
        try:
            {"gemm": gemm_cases, "attn": attn_cases, "bw": bw_cases}[w]()
        except Exception as ex:  # noqa: BLE001
            print("case group failed:", w, repr(ex), flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/perf_ops.json", "w"), indent=1)
