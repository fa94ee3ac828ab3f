"""GPU: A/B of the start order of causal attention workgroups -- id order (slam_attn_set_fwd_qf 50) against heaviest sequence block
first inside every XCD's run (51, round 5) -- at the Llama shapes of the bench workloads (C3: B 31, T 380, 32 q / 8 kv heads, D 128;
C2: B 8; C1: TinyLlama 32 / 4 heads, D 64, B 1, T ~170) and one long-sequence shape.  Interleaved, HIP events, median of 7 rounds of
5 launches.  Prints one JSON object."""
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


def shape(B, T, Hq, Hkv, D):
    qkv = torch.randn(B * T, (Hq + 2 * Hkv) * D, device=dev).to(torch.bfloat16)
    q2, k2, v2 = qkv[:, : Hq * D], qkv[:, Hq * D:(Hq + Hkv) * D], qkv[:, (Hq + Hkv) * D:]
    Tp = ops.round_up(T, 64)
    km = torch.zeros((B, Tp), dtype=torch.uint8, device=dev)
    km[:, :T] = 1
    cos, sin = (t.to(dev) for t in rope_tables(T, D, 500000.0))
    scale = D ** -0.5
    o, lse = ops.attn_fwd(q2, k2, v2, B, T, Hq, Hkv, D, True, scale, key_mask=km)
    do = torch.randn(B * T, Hq * D, device=dev).to(torch.bfloat16)
    dqkv = torch.empty_like(qkv)

    def fwd():
        ops.attn_fwd(q2, k2, v2, B, T, Hq, Hkv, D, True, scale, key_mask=km, out=o)

    def bwd():
        ops.attn_bwd(q2, k2, v2, o, do, lse, dqkv[:, : Hq * D], dqkv[:, Hq * D:(Hq + Hkv) * D],
                     dqkv[:, (Hq + Hkv) * D:], B, T, Hq, Hkv, D, True, scale, key_mask=km, rope=(cos, sin))
    return {"fwd": fwd, "bwd(dq+dkdv)": bwd}


def main():
    out = {}
    for name, dims in (("C3 31x380 32/8 D128", (31, 380, 32, 8, 128)), ("C2 8x380 32/8 D128", (8, 380, 32, 8, 128)),
                       ("C1 1x170 32/4 D64", (1, 170, 32, 4, 64)), ("long 4x2048 32/8 D128", (4, 2048, 32, 8, 128))):
        fns = shape(*dims)
        res = {k: {50: [], 51: []} for k in fns}
        for rnd in range(8):
            for knob in (50, 51):
                call("slam_attn_set_fwd_qf", knob)
                for k, fn in fns.items():
                    fn()
                    torch.cuda.synchronize()
                    t = timed(fn)
                    if rnd:
                        res[k][knob].append(t)
        call("slam_attn_set_fwd_qf", 51)
        out[name] = {k: {"id_order_us": round(statistics.median(v[50]), 1), "heaviest_first_us": round(statistics.median(v[51]), 1),
                         "ratio": round(statistics.median(v[51]) / statistics.median(v[50]), 4)} for k, v in res.items()}
        print(name, out[name], file=sys.stderr, flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
