"""GPU: the tall-skinny LoRA kernels at the C3 shapes (M = 31 x 380 = 11780): first hop u = x A^T (lora_a_fwd), the narrow dL/du product of
the backward (dy [M, 6144] x B_ext [64, 6144]^T), the gram products dA = du^T x and dB = dy^T u.  HIP events, median of 7 rounds of 10
launches; GB/s = the bytes of the tall operand / time.  `python tools/lora_bench.py`"""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from slam_llm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, n=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def main():
    M, K, N, R = 11780, 4096, 6144, int(sys.argv[1]) if len(sys.argv) > 1 else 32
    x = torch.randn(M, K + 64, device=dev).to(torch.bfloat16)
    a = (torch.randn(R, K, device=dev) * 0.02).to(torch.bfloat16)
    dy = torch.randn(M, N, device=dev).to(torch.bfloat16)
    bext = (torch.randn(64, N, device=dev) * 0.02).to(torch.bfloat16)
    du = torch.empty(M, 64, dtype=torch.bfloat16, device=dev)
    da = torch.zeros(R, K, device=dev)
    db = torch.zeros(4096, R // 2, device=dev)
    u = x[:, K: K + R]
    cases = {
        "lora_a_fwd u = x A^T": (lambda: ops.lora_a_fwd(x[:, :K], a, u), M * K * 2),
        "narrow dL/du via gemm_nt": (lambda: ops.gemm_nt(dy, bext, out=du), M * N * 2),
        "narrow dL/du via lora_a_fwd": (lambda: ops.lora_a_fwd(dy, bext, du), M * N * 2),
        "gram dA = du^T x": (lambda: ops.skinny_gram(du[:, :R], x[:, :K], da, K, 1), M * K * 2),
        "gram dB = dy_q^T u": (lambda: ops.skinny_gram(u[:, : R // 2], dy[:, :4096], db, 1, R // 2), M * 4096 * 2),
    }
    out = {}
    for name, (fn, nbytes) in cases.items():
        ts = [timed(fn) for _ in range(7)]
        t = statistics.median(ts)
        out[name] = dict(us=round(t, 1), GBps=round(nbytes / t / 1e3, 0))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
