"""Helpers for reading tests/golden/*.npz (written by oracle/make_golden.py from the reference itself)."""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def sub(fx, name, arr):
    """subsample `arr` the way the fixture `name` was packed; returns (golden_subsample, arr_subsample)"""
    a = np.asarray(arr, dtype=np.float32).reshape(-1)
    stride = int(fx[name + ".__stride"])
    return fx[name], a[::stride]


def check_packed(fx, name, arr, atol, rtol, norm_rtol=None):
    g, a = sub(fx, name, arr)
    assert g.shape == a.shape, f"{name}: shape {a.shape} vs golden {g.shape}"
    err = np.abs(g - a)
    tol = atol + rtol * np.abs(g)
    assert (err <= tol).all(), f"{name}: max err {err.max():.3e} (tol {tol[err.argmax()]:.3e}) at {err.argmax()}"
    if norm_rtol is not None:
        n = float(np.sqrt((np.asarray(arr, dtype=np.float64) ** 2).sum()))
        gn = float(fx[name + ".__norm"])
        assert abs(n - gn) <= norm_rtol * max(gn, 1e-12), f"{name}: norm {n} vs golden {gn}"


# Gradient-cosine floors of the model-level GPU tests.  SURVEY 8(c) states >= 0.999 for the frozen-encoder path (LoRA + projector
# gradients): kept.  Every other entry is a stated DEVIATION (DESIGN.md section 7 lists them with the reasons), set to at least 2x the
# worst deviation measured on MI355X over the whole suite (tools/margins_report.py -> profiles/r05_margins.md): VERDICT r4 next #5b / #5c.
FLOORS = dict(
    frozen=0.999,               # worst measured 1 - cos: 3.4e-4 (projector linear1, toy widths); 6.7e-5 at the headline geometry
    c1_full_depth=0.9965,       # 1.59e-3: layer 21 q_proj lora_A under 22 layers of bf16 residual stream (dQ is a cancelling sum over keys)
    unfrozen_fixture=0.9975,    # 9.8e-4 / 1.2e-3: Whisper blocks.0 query / key weights at toy widths vs the reference fixture
    unfrozen=0.996,             # 1.96e-3 (HuBERT-base pos_conv weight_g), 1.5e-3 (k_proj) at toy widths; 1.2e-3 Whisper + cov1d / Q-Former
    unfrozen_fe=0.993,          # 3.5e-3: conv_layers.0 under seven bf16 conv / LayerNorm adjoints at 64-channel widths
    unfrozen_wavlm_base=0.989,  # 5.2e-3: WavLM Base at toy widths, the noisiest member of the family
    unfrozen_gate=0.983,        # 8.2e-3: grep_linear.weight / relative_attention_bias (cancelling sums of dS)
)

# the deviation (1 - cosine) measured on MI355X behind each relaxed floor: _check_grads() warns (drift alarm, never a failure) when the worst
# tensor of a run leaves [0.5, 1.5] x this value -- a 1.9x regression no longer passes silently under a 2x-wide floor
EXPECT = dict(
    frozen=3.4e-4, c1_full_depth=1.59e-3, c3_full_depth=5.6e-3, unfrozen_fixture=1.2e-3, unfrozen=1.96e-3, unfrozen_fe=3.5e-3,
    unfrozen_wavlm_base=5.2e-3, unfrozen_gate=8.2e-3,
)


def drift_report():
    """per floor family of FLOORS: the worst deviation this run measured against the expected one; a list of (family, measured, expected,
    inside the [0.5, 1.5] band?) -- tests/conftest.py prints it at the end of a GPU run and warns about every family outside its band"""
    by_allowed = {round(1.0 - v, 9): k for k, v in FLOORS.items()}
    worst = {}
    for _t, _what, measured, allowed in MARGINS:
        fam = by_allowed.get(round(allowed, 9))
        if fam is not None:
            worst[fam] = max(worst.get(fam, 0.0), measured)
    return [(fam, m, EXPECT[fam], 0.5 * EXPECT[fam] <= m <= 1.5 * EXPECT[fam]) for fam, m in sorted(worst.items()) if fam in EXPECT]

MARGINS = []      # (test id, what, measured deviation, allowed deviation): written by tests/conftest.py when SLAM_TEST_MARGINS is set


def floor_check(cs, floor, what=""):
    """assert a cosine against its floor AND record how much of the allowed deviation (1 - floor) the measured one (1 - cs) uses: the
    suite-wide rule of VERDICT r4 next #5c -- no bound within 2x of a measured value -- is checked from these records
    (tools/margins_report.py, profiles/r05_margins.md)"""
    import os
    cs, floor = float(cs), float(floor)
    MARGINS.append((os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], str(what)[:120], 1.0 - cs, 1.0 - floor))
    assert cs >= floor, f"{what}: cosine {cs} below the floor {floor}"


def drift_check(what, measured, expected, lo=0.5, hi=1.5):
    """warn (never fail) when `measured` leaves [lo, hi] x `expected`; returns True when inside"""
    import warnings
    ok = lo * expected <= measured <= hi * expected
    if not ok:
        warnings.warn(f"DRIFT {what}: measured deviation {measured:.3e} vs expected {expected:.3e} (alarm band x{lo}..x{hi})")
    return ok


def check_logits(hip_logits, ref_logits, attention_mask, labels, what, atol=6e-2, rtol=2e-2, other_stride=5, min_top1=None):
    """VERDICT r5 missing #1 / SURVEY 8(c) "logits compared on a sampled subset": the every-row eval forward's logits
    (`outputs.logits`, [B, T, V] bf16 on the device) against the oracle's ([B, T, V] fp32, CPU) on EVERY row that carries a label
    (shifted: row t predicts labels[t + 1]) plus every `other_stride`-th other non-pad row, all V columns:
      * |hip - ref| <= atol + rtol * max|ref|  (the bf16 bound the reference fixtures use, tests/test_model_gpu.py:63-71);
      * top-1 on the label rows: the oracle's arg-max token must be HIP's arg-max or within the same bound of it (two near-tied logits
        may swap under bf16), and the plain agreement rate is returned (asserted >= min_top1 when given).
    Pad rows are garbage by design and skipped.  Returns a dict of the measured numbers."""
    import torch
    B, T, V = ref_logits.shape
    assert tuple(hip_logits.shape) == (B, T, V), (tuple(hip_logits.shape), (B, T, V))
    am = attention_mask.bool().reshape(-1).cpu()
    lab = torch.nn.functional.pad(labels, (0, 1), value=-100)[:, 1:].reshape(-1).cpu() != -100
    lab &= am
    other = am & ~lab
    idx_other = torch.nonzero(other).flatten()[::other_stride]
    idx_lab = torch.nonzero(lab).flatten()
    rows = torch.cat([idx_lab, idx_other])
    dev = hip_logits.device
    ref = ref_logits.reshape(B * T, V)[rows].to(dev, torch.float32)
    got = hip_logits.reshape(B * T, V)[rows.to(dev)].float()
    assert bool(torch.isfinite(got).all()), f"{what}: non-finite logits on valid rows"
    err = (got - ref).abs()
    bound = atol + rtol * float(ref.abs().max())
    worst = float(err.max())
    rms = float((err.double() ** 2).mean().sqrt())
    assert worst <= bound, f"{what}: logits max |err| {worst:.4f} > {bound:.4f} (rms {rms:.4f}) over {rows.numel()} rows x {V}"
    nl = idx_lab.numel()
    stats = dict(rows=int(rows.numel()), label_rows=int(nl), max_err=worst, rms_err=rms, bound=bound, ref_absmax=float(ref.abs().max()))
    if nl:
        r_top = ref[:nl].argmax(-1)
        g_top = got[:nl].argmax(-1)
        agree = float((r_top == g_top).float().mean())
        # the oracle's winner must be within the bound of HIP's maximum (a swap of two near-tied logits is not an error)
        gap = got[:nl].max(-1).values - got[:nl].gather(1, r_top[:, None]).squeeze(1)
        assert float(gap.max()) <= 2 * bound, f"{what}: oracle's top-1 token sits {float(gap.max()):.4f} below HIP's maximum on a label row"
        stats.update(top1_agree=agree, top1_gap_max=float(gap.max()))
        if min_top1 is not None:
            assert agree >= min_top1, f"{what}: top-1 agreement {agree:.4f} < {min_top1}"
    print(f"logits {what}: {stats}")
    rep = os.environ.get("SLAM_TEST_REPORT")
    if rep:
        with open(rep + ".logits.tsv", "a") as f:
            f.write(f"{what}\t" + "\t".join(f"{k}={v:.5g}" if isinstance(v, float) else f"{k}={v}" for k, v in stats.items()) + "\n")
    return stats


def cosine(a, b):
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    return float((a * b).sum() / (np.sqrt((a * a).sum() * (b * b).sum()) + 1e-30))


def mix64(seed: int, q):
    """host restatement of the mask hash of csrc/common.h: slam_mix64(seed ^ (q * 0xD1342543DE82EF95)) -- the splitmix64 finaliser over the
    group index q = element index >> 2; q = uint64 array"""
    q = np.asarray(q, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed & (2 ** 64 - 1)) ^ (q * np.uint64(0xD1342543DE82EF95))
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def attn_keep_mask(seed: int, p: float, B: int, H: int, Tq: int, Tk: int, Tqp: int, Tkp: int):
    """host restatement of the attention kernels' counter-based dropout mask (csrc/attention.hip attn_keep4 over
    csrc/common.h slam_mix64): element (b, h, q, k) lives at index ((b*H + h)*Tqp + q)*Tkp + k; one splitmix64 word per group of
    4 consecutive indices (mix64 below), 16 bits each, keep = bits >= round(p * 65536).  Returns float32 [B, H, Tq, Tk] of 0 / 1."""
    import numpy as np
    thresh = min(65535, int(p * 65536.0 + 0.5))
    b, h, q, k = np.meshgrid(np.arange(B, dtype=np.uint64), np.arange(H, dtype=np.uint64), np.arange(Tq, dtype=np.uint64),
                             np.arange(Tk, dtype=np.uint64), indexing="ij")
    idx = ((b * np.uint64(H) + h) * np.uint64(Tqp) + q) * np.uint64(Tkp) + k
    z = mix64(seed, idx >> np.uint64(2))
    bits = (z >> (np.uint64(16) * (idx & np.uint64(3)))) & np.uint64(0xFFFF)
    return (bits >= np.uint64(thresh)).astype(np.float32)


def wavlm_train_masks(fx, tag, n_layers, p=0.1):
    """the recorded regulariser masks of tests/golden/wavlm_train_tiny.npz, case `tag`, as oracle.wavlm_encoder's `train` argument"""
    import torch
    kept = [bool(k) for k in fx[tag + ".kept"]]

    def m(name):
        shape = [int(x) for x in fx[f"{tag}.mask.{name}.shape"]]
        bits = np.unpackbits(fx[f"{tag}.mask.{name}"])[: int(np.prod(shape))]
        return torch.from_numpy(bits.reshape(shape).astype(np.float32)) / (1.0 - p)
    tr = {"input": m("input"), "x": m("x"), "layers": []}
    for i in range(n_layers):
        tr["layers"].append({k: m(f"l{i}.{k}") for k in ("attn", "d1", "d2", "d3")} if kept[i] else None)
    return tr


def layerdrop_seed(pattern, layerdrop):
    """numpy seed whose first len(pattern) draws keep (np.random.random() > layerdrop, WavLM.py:596-597) / skip exactly as `pattern`"""
    for s in range(1000):
        np.random.seed(s)
        if tuple(bool(np.random.random() > layerdrop) for _ in pattern) == tuple(pattern):
            return s
    raise RuntimeError("no seed")
