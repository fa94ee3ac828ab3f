"""GPU: attention kernel timings of ONE build of the library (SLAM_HIP_LIB selects it: tools/r05_call7.sh runs the round-4/5 baseline
build and the current one alternately) at the shapes of the C3 step: Whisper-large-v3 encoder forward (31 x 1500 frames, 20 heads, D 64,
no LSE = the frozen encoder's launch; with LSE = the un-frozen encoder's), Llama-3 forward and backward (31 x 380, 32 / 8 heads, D 128).
HIP events, median of 7 rounds of 5 launches.  Prints one JSON object."""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from slam_llm_amd import ops  # noqa: E402
from slam_llm_amd.host_tables import rope_tables  # noqa: E402
from slam_llm_amd.lib import LIB_PATH, call  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, n=5):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3   # us


def main():
    fns = {}
    B, T, H, D = 31, 1500, 20, 64
    qkv = torch.randn(B * T, 3 * H * D, device=dev).to(torch.bfloat16)
    q2, k2, v2 = qkv[:, : H * D], qkv[:, H * D: 2 * H * D], qkv[:, 2 * H * D:]
    o = torch.empty(B * T, H * D, device=dev, dtype=torch.bfloat16)
    pre = hasattr(ops, "qscale")   # (builds from the folded-scale form on: the frozen encoder's launch hands over a pre-scaled Q)
    if pre:
        qkv[:, : H * D] = (qkv[:, : H * D].float() * ops.qscale(D ** -0.5)).to(torch.bfloat16)
    fns["whisper_fwd_no_lse"] = ((lambda: ops.attn_fwd(q2, k2, v2, B, T, H, H, D, False, D ** -0.5, want_lse=False, out=o, q_prescaled=True)) if pre
                                 else (lambda: ops.attn_fwd(q2, k2, v2, B, T, H, H, D, False, D ** -0.5, want_lse=False, out=o)))
    fns["whisper_fwd_lse"] = lambda: ops.attn_fwd(q2, k2, v2, B, T, H, H, D, False, D ** -0.5, want_lse=True, out=o)
    Bl, Tl, Hq, Hkv, Dl = 31, 380, 32, 8, 128
    qkvl = torch.randn(Bl * Tl, (Hq + 2 * Hkv) * Dl, device=dev).to(torch.bfloat16)
    ql, kl, vl = qkvl[:, : Hq * Dl], qkvl[:, Hq * Dl:(Hq + Hkv) * Dl], qkvl[:, (Hq + Hkv) * Dl:]
    km = torch.zeros((Bl, ops.round_up(Tl, 64)), dtype=torch.uint8, device=dev)
    km[:, :Tl] = 1
    cos, sin = (t.to(dev) for t in rope_tables(Tl, Dl, 500000.0))
    ol, lsel = ops.attn_fwd(ql, kl, vl, Bl, Tl, Hq, Hkv, Dl, True, Dl ** -0.5, key_mask=km)
    dol = torch.randn(Bl * Tl, Hq * Dl, device=dev).to(torch.bfloat16)
    dqkv = torch.empty_like(qkvl)
    fns["llama_fwd"] = lambda: ops.attn_fwd(ql, kl, vl, Bl, Tl, Hq, Hkv, Dl, True, Dl ** -0.5, key_mask=km, out=ol)
    fns["llama_bwd"] = lambda: ops.attn_bwd(ql, kl, vl, ol, dol, lsel, dqkv[:, : Hq * Dl], dqkv[:, Hq * Dl:(Hq + Hkv) * Dl],
                                            dqkv[:, (Hq + Hkv) * Dl:], Bl, Tl, Hq, Hkv, Dl, True, Dl ** -0.5, key_mask=km, rope=(cos, sin))
    knobs = [None]
    try:
        call("slam_attn_set_fwd_qf", 61)
        knobs = [61, 60]
    except Exception:  # noqa: BLE001  (the baseline build has no such knob)
        pass
    res = {}
    for rnd in range(8):
        for name, fn in fns.items():
            for kn in (knobs if name == "whisper_fwd_no_lse" else knobs[:1]):
                if kn is not None:
                    call("slam_attn_set_fwd_qf", kn)
                fn()
                torch.cuda.synchronize()
                t = timed(fn)
                if rnd:
                    res.setdefault(name + (f"@{kn}" if kn == 60 else ""), []).append(t)
        if knobs[0] is not None:
            call("slam_attn_set_fwd_qf", 61)
    print(json.dumps({"lib": os.path.basename(LIB_PATH), **{k: round(statistics.median(v), 1) for k, v in res.items()}}))


if __name__ == "__main__":
    main()
