"""GPU (tools): the round-4 K-sliced GEMM forms -- (a) mid-M products (two-launch split-K of the 4-wave kernel) and (b) N <= 64 products
(tall-skinny kernel) -- against what the auto rule launched before, results checked against an fp32 matmul of the same bf16 operands.
Interleaved rounds, HIP events, median.  python tools/gemm_midm_bench.py > gpurun_out/gemm_midm.jsonl"""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from slam_llm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
MID = [(11780, 6144, 4160), (11780, 14336, 4096), (11780, 4096, 4096), (672, 4096, 4096), (672, 4096, 11008), (672, 4096, 12288), (672, 4096, 22016), (672, 12288, 4160), (672, 11008, 4096), (672, 22016, 4096),
       (380, 2048, 2048), (380, 2048, 5632), (380, 5632, 2048), (3040, 4096, 4096), (3040, 4096, 14336), (1520, 4096, 4096), (1520, 4096, 14336)]
TS = [(11780, 64, 6144), (672, 64, 12288), (3040, 64, 6144), (380, 64, 2560), (11780, 16, 4096), (11780, 64, 4096)]


def timed(fn, n=3):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def check(a, b, c):
    rows = torch.arange(0, a.shape[0], max(1, a.shape[0] // 97), device=dev)
    ref = a[rows].float() @ b.float().t()
    return float(((c[rows].float() - ref).abs() / (1 + ref.abs())).max())


def main():
    a0 = torch.randn(64, 256, device=dev).to(torch.bfloat16)
    ops.gemm_nt(a0, a0)       # registers the workspace
    for M, N, K in MID:
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        b = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
        c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        modes = {"rule_r3 (cfg0, sk2 off)": [360, 0], "rule_r4 (cfg0, sk2 auto)": [361, 0], "w4 unsplit": [360, 301, 12]}
        for S in (2, 3, 4, 5, 6, 8):
            if (K // 64) // S >= 4:
                modes[f"w4 two-launch S{S}"] = [362, 300 + S, 12]
        res, err = {k: [] for k in modes}, {}

        def run(seq):
            for cfg in seq:
                ops.gemm_set_config(cfg)
            t = timed(lambda: ops.gemm_nt(a, b, out=c))
            ops.gemm_set_config(301); ops.gemm_set_config(361); ops.gemm_set_config(0)
            return t
        for k, seq in modes.items():
            c.zero_()
            run(seq)
            err[k] = round(check(a, b, c), 5)
        for _ in range(5):
            for k, seq in modes.items():
                res[k].append(run(seq))
        med = {k: round(statistics.median(v), 1) for k, v in res.items()}
        print(json.dumps(dict(kind="mid-M", M=M, N=N, K=K, rule_r4=ops.gemm_kernel_name(M, N, K), us=med, rel_err=err)), flush=True)
    for M, N, K in TS:
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        b = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
        res, err = {"tile128x64": [], "tall-skinny": []}, {}
        for name, cfg in (("tile128x64", 370), ("tall-skinny", 371)):
            ops.gemm_set_config(cfg)
            c = ops.gemm_nt(a, b)
            err[name] = round(check(a, b, c), 5)
            # accumulate into fp32 and bf16 outputs, alpha
            cf = torch.ones(M, N, dtype=torch.float32, device=dev)
            ops.gemm_nt(a, b, out=cf, accumulate=True, alpha=0.5)
            rows = torch.arange(0, M, max(1, M // 97), device=dev)
            ref = 1.0 + 0.5 * (a[rows].float() @ b.float().t())
            err[name + " f32 acc"] = round(float(((cf[rows] - ref).abs() / (1 + ref.abs())).max()), 6)
        for _ in range(5):
            for name, cfg in (("tile128x64", 370), ("tall-skinny", 371)):
                ops.gemm_set_config(cfg)
                c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
                res[name].append(timed(lambda: ops.gemm_nt(a, b, out=c)))
        ops.gemm_set_config(371)
        print(json.dumps(dict(kind="tall-skinny", M=M, N=N, K=K, us={k: round(statistics.median(v), 1) for k, v in res.items()}, err=err,
                              GBs=round(2.0 * M * K / statistics.median(res["tall-skinny"]) / 1e3, 1))), flush=True)


if __name__ == "__main__":
    main()
