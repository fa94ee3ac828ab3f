"""GPU: the split-K tail of the 4-wave GEMM on the step's product shapes -- unsplit vs the auto plan vs forced slice counts,
next to whatever the auto RULE picks today (cfg 0).  Interleaved rounds, HIP events, median.  One JSON line per shape.

    python tools/gemm_splitk_sweep.py [c3|c2|c4|all] > gpurun_out/gemm_splitk.jsonl"""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from slam_llm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = {
    # C3 (31 x 380 = 11780 rows): forward products, their dX twins, lm_head chunks
    "c3": [(11780, 6144, 4160), (11780, 4096, 4096), (11780, 28672, 4096), (11780, 4096, 14336), (11780, 14336, 4096),
           (11780, 4096, 28672), (11780, 4096, 6144), (4096, 128256, 4096), (3588, 128256, 4096), (4096, 4096, 128256)],
    # C2 (8 x 380 = 3040 rows)
    "c2": [(3040, 6144, 4160), (3040, 4096, 4096), (3040, 28672, 4096), (3040, 4096, 14336), (3040, 14336, 4096), (3040, 128256, 4096)],
    # C4 (6 x 112 = 672 rows, Vicuna-7B MHA, r32 on q,v)
    "c4": [(672, 12288, 4160), (672, 4096, 4096), (672, 22016, 4096), (672, 4096, 11008), (672, 11008, 4096), (672, 32000, 4096),
           (672, 4096, 12288), (672, 4096, 22016)],
}


def timed(fn, n=3):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    shapes = sum(SHAPES.values(), []) if which == "all" else SHAPES[which]
    for M, N, K in shapes:
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        b = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
        c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        tiles = -(-M // 256) * -(-N // 256)
        modes = {"rule(cfg0,nosplit)": (0, 301), "w4_off": (12, 301), "w4_auto": (12, 300)}
        for S in (2, 3, 4, 5, 6, 8):
            if (K // 64) // S >= 4:
                modes[f"w4_S{S}"] = (12, 300 + S)
        res = {k: [] for k in modes}

        def run(cfg, sk):
            ops.gemm_set_config(cfg)
            ops.gemm_set_config(sk)
            return timed(lambda: ops.gemm_nt(a, b, out=c))
        try:
            for k, (cfg, sk) in modes.items():
                run(cfg, sk)
            for _ in range(5):
                for k, (cfg, sk) in modes.items():
                    res[k].append(run(cfg, sk))
        finally:
            ops.gemm_set_config(301)
            ops.gemm_set_config(0)
        med = {k: round(statistics.median(v), 1) for k, v in res.items()}
        best = min(med, key=med.get)
        print(json.dumps(dict(M=M, N=N, K=K, tiles=tiles, tail=tiles % 256, rule_kernel=ops.gemm_kernel_name(M, N, K), us=med, best=best,
                              TF_best=round(2.0 * M * N * K / med[best] / 1e6, 1), TF_off=round(2.0 * M * N * K / med["w4_off"] / 1e6, 1))), flush=True)
        del a, b, c


if __name__ == "__main__":
    main()
