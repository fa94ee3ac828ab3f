"""GPU (tools): transposed-read attention kernels (slam_attn_set_fwd_qf 41, shipped) against the round-3 kernels on the [B,H,D,Tp] copies (40):
outputs compared (max abs difference per tensor), times interleaved (HIP events, median of 7 rounds x 5 launches), at the C3 shapes.
python tools/attn_tr_ab.py > gpurun_out/attn_tr_ab.json"""
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
TR = [True]     # forward: row-major V (transposed-read kernel) | the [B,H,D,Tp] copy (round-3 kernel)


def timed(fn, n=5):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def llama(B=31, T=380, Hq=32, Hkv=8, D=128, causal=True, mask=True):
    qkv = torch.randn(B * T, (Hq + 2 * Hkv) * D, device=dev).to(torch.bfloat16)
    q2, k2, v2 = qkv[:, : Hq * D], qkv[:, Hq * D:(Hq + Hkv) * D], qkv[:, (Hq + Hkv) * D:]
    qt = ops.head_rope_transpose(qkv, 0, B, T, Hq, D)
    kt = ops.head_rope_transpose(qkv, Hq * D, B, T, Hkv, D)
    vt = ops.head_rope_transpose(qkv, (Hq + Hkv) * D, B, T, Hkv, D)
    km = None
    if mask:
        km = torch.zeros((B, vt.shape[-1]), dtype=torch.uint8, device=dev)
        km[:, :T] = 1
    cos, sin = (t.to(dev) for t in rope_tables(T, D, 500000.0))
    scale = D ** -0.5
    o, lse = ops.attn_fwd(q2, k2, vt, B, T, Hq, Hkv, D, causal, scale, key_mask=km)
    do = torch.randn(B * T, Hq * D, device=dev).to(torch.bfloat16)
    dot = ops.head_rope_transpose(do, 0, B, T, Hq, D)
    outs = {}

    def fwd(tag=None):
        oo, ll = ops.attn_fwd(q2, k2, v2 if TR[0] else vt, B, T, Hq, Hkv, D, causal, scale, key_mask=km)
        if tag is not None:
            outs[tag] = (oo, ll[..., :T].clone())

    def bwd(tag=None):
        dqkv = torch.empty_like(qkv)
        ops.attn_bwd(q2, k2, v2, o, do, lse, dqkv[:, : Hq * D], dqkv[:, Hq * D:(Hq + Hkv) * D],
                     dqkv[:, (Hq + Hkv) * D:], B, T, Hq, Hkv, D, causal, scale, key_mask=km, rope=(cos, sin) if causal else None)
        if tag is not None:
            outs[tag] = (dqkv,)
    return fwd, bwd, outs


def whisper(B=31, T=1500, H=20, D=64):
    qkv = torch.randn(B * T, 3 * H * D, device=dev).to(torch.bfloat16)
    q2, k2 = qkv[:, : H * D], qkv[:, H * D: 2 * H * D]
    vt = ops.head_rope_transpose(qkv, 2 * H * D, B, T, H, D)
    o = torch.empty((B * T, H * D), dtype=torch.bfloat16, device=dev)
    outs = {}

    def fwd(tag=None):
        ops.attn_fwd(q2, k2, qkv[:, 2 * H * D:] if TR[0] else vt, B, T, H, H, D, False, D ** -0.5, want_lse=False, out=o)
        if tag is not None:
            outs[tag] = (o.clone(),)
    return fwd, None, outs


def main():
    res = {}
    cases = {"llama(B31,T380,32q/8kv,D128,causal)": llama(), "whisper_enc_bwd(B4,T1500,20h,D64,bidirectional)": llama(4, 1500, 20, 20, 64, False, False),
             "whisper_fwd(B31,T1500,20h,D64)": whisper()}
    for name, (fwd, bwd, outs) in cases.items():
        for which, fn in (("fwd", fwd), ("bwd", bwd)):
            if fn is None:
                continue
            for knob, tag in ((40, "copy"), (41, "tr")):
                call("slam_attn_set_fwd_qf", knob)
                TR[0] = knob == 41
                fn(tag)
            torch.cuda.synchronize()
            diff = [float((a.float() - b.float()).abs().max()) for a, b in zip(outs["copy"], outs["tr"])]
            t = {"copy": [], "tr": []}
            for _ in range(7):
                for knob, tag in ((40, "copy"), (41, "tr")):
                    call("slam_attn_set_fwd_qf", knob)
                    TR[0] = knob == 41
                    t[tag].append(timed(fn))
            res[f"{name} {which}"] = dict(max_abs_diff=diff, copy_us=round(statistics.median(t["copy"]), 1), tr_us=round(statistics.median(t["tr"]), 1))
    call("slam_attn_set_fwd_qf", 41)
    json.dump(res, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
