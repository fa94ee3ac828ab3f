"""GPU: streaming rate of the skinny (decode) GEMM per Llama-3-8B weight shape, split-K sweep.  Weights are rotated
through > 1 GB of distinct copies so that neither L2 nor the 256 MB Infinity Cache can serve them."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from slam_llm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16
shapes = [("qkv+lora", 6144, 4160), ("o", 4096, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336),
          ("lm_head", 128256, 4096), ("lora_A", 32, 4096)]
res = []
for name, N, K in shapes:
    nbytes = N * K * 2
    copies = max(2, min(64, (1 << 30) // nbytes + 1))
    Ws = [torch.randn(N, K, device=dev).to(torch.bfloat16) for _ in range(copies)]
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    line = {"shape": name, "M": M, "N": N, "K": K}
    for splits in (0, 1, 2, 4, 8, 16, 32):
        ops.SKINNY_SPLITS = splits
        for w in Ws[:2]:
            ops.gemm_nt(x, w, out=out)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = max(1, 64 // copies)
        s.record()
        for _ in range(reps):
            for w in Ws:
                ops.gemm_nt(x, w, out=out)
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3 / (reps * copies)
        line[f"S{splits}"] = f"{us:.1f}us {nbytes / us / 1e3:.0f}GB/s"
    ops.SKINNY_SPLITS = 0
    print(line, flush=True)
    res.append(line)
    del Ws
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/skinny_bench.json", "w"), indent=1)
