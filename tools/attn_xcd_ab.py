"""GPU: A/B of the attention kernels' workgroup numbering at the C3 shapes -- hardware round-robin order (slam_attn_set_fwd_qf 20)
against the XCD-aware order (21, shipped).  Whisper encoder forward (B 31, T 1500, 20 heads, D 64, bidirectional) and the Llama
forward / backward (B 31, T 380, 32 q / 8 kv heads, D 128, causal, fused RoPE gradient), interleaved, HIP events, median of 7
rounds of 5 launches.  Prints one JSON object; `python tools/attn_xcd_ab.py > gpurun_out/attn_xcd_ab.json`."""
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


def llama():
    B, T, Hq, Hkv, D = 31, 380, 32, 8, 128
    qkv = torch.randn(B * T, (Hq + 2 * Hkv) * D, device=dev).to(torch.bfloat16)
    q2, k2, v2 = qkv[:, : Hq * D], qkv[:, Hq * D:(Hq + Hkv) * D], qkv[:, (Hq + Hkv) * D:]
    qt = ops.head_rope_transpose(qkv, 0, B, T, Hq, D)
    kt = ops.head_rope_transpose(qkv, Hq * D, B, T, Hkv, D)
    vt = ops.head_rope_transpose(qkv, (Hq + Hkv) * D, B, T, Hkv, D)
    km = torch.zeros((B, vt.shape[-1]), dtype=torch.uint8, device=dev)
    km[:, :T] = 1
    cos, sin = (t.to(dev) for t in rope_tables(T, D, 500000.0))
    scale = D ** -0.5
    o, lse = ops.attn_fwd(q2, k2, vt, B, T, Hq, Hkv, D, True, scale, key_mask=km)
    do = torch.randn(B * T, Hq * D, device=dev).to(torch.bfloat16)
    dot = ops.head_rope_transpose(do, 0, B, T, Hq, D)
    dqkv = torch.empty_like(qkv)

    def fwd():
        ops.attn_fwd(q2, k2, vt, B, T, Hq, Hkv, D, True, scale, key_mask=km, out=o)

    def bwd():
        ops.attn_bwd(q2, k2, v2, o, do, lse, dqkv[:, : Hq * D], dqkv[:, Hq * D:(Hq + Hkv) * D],
                     dqkv[:, (Hq + Hkv) * D:], B, T, Hq, Hkv, D, True, scale, key_mask=km, rope=(cos, sin))
    return {"llama_fwd": fwd, "llama_bwd(dq+dkdv)": bwd}


def whisper():
    B, T, H, D = 31, 1500, 20, 64
    qkv = torch.randn(B * T, 3 * H * D, device=dev).to(torch.bfloat16)
    q2, k2 = qkv[:, : H * D], qkv[:, H * D: 2 * H * D]
    vt = ops.head_rope_transpose(qkv, 2 * H * D, B, T, H, D)
    o = torch.empty((B * T, H * D), dtype=torch.bfloat16, device=dev)

    def fwd():
        ops.attn_fwd(q2, k2, vt, B, T, H, H, D, False, D ** -0.5, want_lse=False, out=o)
    return {"whisper_fwd": fwd}


def main():
    only = sys.argv[2] if len(sys.argv) > 2 and sys.argv[1] == "--only" else None
    if only:   # (before ANY attention launch, the set-up ones included: the PMC pass averages over every launch of a kernel)
        call("slam_attn_set_fwd_qf", 20 if only == "hw" else 21)
    fns = {}
    fns.update(whisper())
    fns.update(llama())
    if only:   # one order only, 3 launches each: the run profiled with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
        for fn in fns.values():
            for _ in range(3):
                fn()
        torch.cuda.synchronize()
        call("slam_attn_set_fwd_qf", 21)
        return
    res = {k: {"hw_order_us": [], "xcd_order_us": []} for k in fns}
    for name, fn in fns.items():
        for knob in (20, 21):
            call("slam_attn_set_fwd_qf", knob)
            timed(fn, 3)
        for _ in range(7):
            for knob, key in ((20, "hw_order_us"), (21, "xcd_order_us")):
                call("slam_attn_set_fwd_qf", knob)
                res[name][key].append(timed(fn))
    call("slam_attn_set_fwd_qf", 21)
    out = {}
    for k, v in res.items():
        a, b = statistics.median(v["hw_order_us"]), statistics.median(v["xcd_order_us"])
        out[k] = dict(hw_order_us=round(a, 1), xcd_order_us=round(b, 1), speedup=round(a / b, 3),
                      hw_all=[round(x, 1) for x in v["hw_order_us"]], xcd_all=[round(x, 1) for x in v["xcd_order_us"]])
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
