"""GPU: A/B of the D = 128 dK / dV kernel forms -- 8 waves x 16 keys (slam_attn_set_fwd_qf 70) against 4 waves x 32 keys (71, round 5) -- at the
Llama shapes of the bench workloads (C3: B 31, T 380, 32 q / 8 kv heads; C2: B 8; C4: Vicuna MHA 32 / 32, B 6, T 112) and a long sequence.
The timed call is slam_attn_bwd (dQ + dK/dV); the dQ kernel is the same in both arms.  Interleaved, HIP events, median of 7 rounds of 5 launches."""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from slam_llm_amd import ops  # noqa: E402
from slam_llm_amd.host_tables import rope_tables  # noqa: E402
from slam_llm_amd.lib import call  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, n=5):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3   # us


def shape(B, T, Hq, Hkv, D=128):
    qkv = torch.randn(B * T, (Hq + 2 * Hkv) * D, device=dev).to(torch.bfloat16)
    q2, k2, v2 = qkv[:, : Hq * D], qkv[:, Hq * D:(Hq + Hkv) * D], qkv[:, (Hq + Hkv) * D:]
    km = torch.zeros((B, ops.round_up(T, 64)), dtype=torch.uint8, device=dev)
    km[:, :T] = 1
    cos, sin = (t.to(dev) for t in rope_tables(T, D, 500000.0))
    scale = D ** -0.5
    o, lse = ops.attn_fwd(q2, k2, v2, B, T, Hq, Hkv, D, True, scale, key_mask=km)
    do = torch.randn(B * T, Hq * D, device=dev).to(torch.bfloat16)
    dqkv = torch.empty_like(qkv)
    return lambda: ops.attn_bwd(q2, k2, v2, o, do, lse, dqkv[:, : Hq * D], dqkv[:, Hq * D:(Hq + Hkv) * D], dqkv[:, (Hq + Hkv) * D:],
                                B, T, Hq, Hkv, D, True, scale, key_mask=km, rope=(cos, sin))


def main():
    out = {}
    for name, dims in (("C3 31x380 32/8", (31, 380, 32, 8)), ("C2 8x380 32/8", (8, 380, 32, 8)), ("C4 6x112 32/32", (6, 112, 32, 32)),
                       ("long 4x2048 32/8", (4, 2048, 32, 8))):
        fn = shape(*dims)
        res = {70: [], 71: []}
        for rnd in range(8):
            for knob in (70, 71):
                call("slam_attn_set_fwd_qf", knob)
                fn()
                torch.cuda.synchronize()
                t = timed(fn)
                if rnd:
                    res[knob].append(t)
        call("slam_attn_set_fwd_qf", 71)
        out[name] = {"16_keys_per_wave_us": round(statistics.median(res[70]), 1), "32_keys_per_wave_us": round(statistics.median(res[71]), 1),
                     "ratio": round(statistics.median(res[71]) / statistics.median(res[70]), 4)}
        print(name, out[name], file=sys.stderr, flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
