"""GPU (tools): second hop of the LoRA backward under lora_dropout -- slam_lora_hop_dropout (one pass over dx) against the two-launch form
(product into a scratch buffer + slam_dropout_bf16(accumulate)) at the C3 / C4 shapes; interleaved, median."""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from slam_llm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, n=5):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for M, K in ((11780, 4096), (672, 4096), (3040, 4096)):
    ext = torch.randn(M, K + 64, device=dev).to(torch.bfloat16)
    a_t = (torch.randn(K, 64, device=dev) * 0.05).to(torch.bfloat16)
    drop = (0.05, 1234, 8 << 40)
    two = lambda: ops.dropout(ops.gemm_nt(ext[:, K:], a_t), *drop, out=ext[:, :K], accumulate=True)   # noqa: E731
    one = lambda: ops.lora_hop_dropout(ext[:, K:], a_t, ext[:, :K], drop)                               # noqa: E731
    two(); one()
    t = {"two launches": [], "fused": []}
    for _ in range(7):
        t["two launches"].append(timed(two))
        t["fused"].append(timed(one))
    print(json.dumps(dict(M=M, K=K, us={k: round(statistics.median(v), 1) for k, v in t.items()},
                          fused_TBs=round(4.0 * M * K / statistics.median(t["fused"]) / 1e6, 2))), flush=True)
