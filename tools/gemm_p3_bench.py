"""GPU: the two-workgroups-per-CU GEMM (cfg 8) against the shipped kernels on the Whisper-large-v3 encoder products of the C3 batch
(M = 31 x 1500), with their epilogues; interleaved rounds, best of 4 x 5 launches.  Usage: python tools/gemm_p3_bench.py [cfgs=7,8,6,12]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_llm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M = 46500
cases = [("qkv  bias", 3840, 1280, dict(bias=True)), ("out  bias+res", 1280, 1280, dict(bias=True, res=True)),
         ("fc1  bias+gelu", 5120, 1280, dict(bias=True, act=ops.ACT_GELU)), ("fc2  bias+res", 1280, 5120, dict(bias=True, res=True)),
         ("proj1 (C3 projector) relu", 2048, 6400, dict(bias=True, act=ops.ACT_RELU, M=9300))]
variants = sys.argv[1].split(",") if len(sys.argv) > 1 else ["7", "8", "8n", "6", "12"]    # "8n" = cfg 8 without the stagger
rows = []
for name, N, K, ep in cases:
    m = ep.get("M", M)
    a = torch.randn(m, K, device=dev).to(torch.bfloat16)
    b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    c = torch.empty(m, N, device=dev, dtype=torch.bfloat16)
    bias = torch.randn(N, device=dev) if ep.get("bias") else None
    res = torch.randn(m, N, device=dev).to(torch.bfloat16) if ep.get("res") else None
    best = {v: 1e9 for v in variants}

    def run():
        ops.gemm_nt(a, b, out=c, bias=bias, residual=res, act=ep.get("act", ops.ACT_NONE))
    for rnd in range(4):
        for v in variants:
            ops.gemm_set_config(380 if v.endswith("n") else 381)
            ops.gemm_set_config(int(v.rstrip("n")))
            run()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5):
                run()
            e.record()
            torch.cuda.synchronize()
            best[v] = min(best[v], s.elapsed_time(e) / 5)
    ops.gemm_set_config(0)
    ops.gemm_set_config(381)
    row = {"case": name, "shape": f"{m}x{N}x{K}", **{f"cfg{v}": {"us": round(best[v] * 1e3, 1), "TF": round(2.0 * m * N * K / (best[v] * 1e-3) / 1e12)} for v in variants}}
    rows.append(row)
    print(json.dumps(row), flush=True)
