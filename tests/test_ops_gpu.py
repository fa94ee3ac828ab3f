"""GPU: every HIP kernel against a plain PyTorch fp32 reference of the same op (through the C ABI).

Tolerances: bf16 outputs carry ~2^-8 relative rounding; accumulations are fp32.  Each check states its bound."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests import golden_util as G  # noqa: E402
pytestmark = pytest.mark.gpu


def _ops():
    from slam_llm_amd import ops
    return ops


def rnd(shape, dev, std=1.0, seed=0, dtype=torch.bfloat16):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * std).to(dtype).to(dev)


def assert_close(got, ref, atol, rtol, what=""):
    got = got.float().cpu()
    ref = ref.float().cpu()
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = err > tol
    assert not bad.any(), (f"{what}: {int(bad.sum())}/{bad.numel()} elements out of tolerance; max err "
                           f"{float(err.max()):.4e} at {int(err.argmax())}, ref there {float(ref.flatten()[err.argmax()]):.4e}")


# ----------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("cfg", [1, 2, 3, 4, 6, 7, 11, 12])
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (300, 200, 64), (1000, 388, 192), (77, 64, 4160), (5000, 4100, 256)])
def test_gemm_plain(dev, cfg, M, N, K):
    ops = _ops()
    a, b = rnd((M, K), dev, seed=1), rnd((N, K), dev, seed=2)
    ops.gemm_set_config(cfg)
    try:
        c = ops.gemm_nt(a, b)
    finally:
        ops.gemm_set_config(0)
    ref = a.float() @ b.float().T
    # fp32 accumulate, one bf16 rounding at the output: |err| <= 2^-8 |ref| + accumulation noise
    assert_close(c, ref, atol=2e-2 * math.sqrt(K / 64), rtol=1e-2, what=f"gemm cfg{cfg} {M}x{N}x{K}")


@pytest.mark.parametrize("cfg", [12, 7, 6, 1])
@pytest.mark.parametrize("M,N,K", [(11780, 4096, 256), (1300, 7 * 256 + 40, 128), (5 * 256, 5 * 256, 192), (300, 33 * 256, 64)])
def test_gemm_raster_group_height_is_a_pure_renumbering(dev, cfg, M, N, K):
    """round 6: the raster group height is chosen per shape class (slam_gemm_set_group_m(0) = the rule of csrc/gemm_bf16.hip:gemm_group_m_for;
    slam_gemm_set_group_m_rule re-parametrises it).  It only changes WHICH workgroup computes a tile: for group heights 1 / 4 / 8 / 12 / 64 and
    the rule itself every output element must be written exactly once (NaN-prefilled output) and be bit-identical to the group-8 order of
    rounds 1-5, on tile grids of 47 x 16, 6 x 8, 5 x 5 and 2 x 33."""
    ops = _ops()
    from slam_llm_amd.lib import call
    a, b = rnd((M, K), dev, seed=11), rnd((N, K), dev, seed=12)
    ops.gemm_set_config(cfg)
    try:
        call("slam_gemm_set_group_m", 8)
        ref = ops.gemm_nt(a, b)
        assert_close(ref, a.float() @ b.float().T, atol=2e-2 * math.sqrt(K / 64), rtol=1e-2, what="reference order")
        for gm in (0, 1, 4, 12, 64):
            call("slam_gemm_set_group_m", gm)
            out = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
            ops.gemm_nt(a, b, out=out)
            assert torch.equal(out, ref), f"group_m {gm}: differs from the reference order"
        call("slam_gemm_set_group_m", 0)
        call("slam_gemm_set_group_m_rule", 3, 5, 7, 2)
        out = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
        ops.gemm_nt(a, b, out=out)
        assert torch.equal(out, ref)

    finally:
        ops.reset_tuning()


@pytest.mark.parametrize("M,N,K", [(700, 1280, 1280), (1500, 520, 1280), (260, 3840, 128), (3000, 5120, 1280)])
def test_gemm_persistent_descriptor_dma_on_strided_views(dev, M, N, K):
    """the auto rule sends K <= 2048 products (the Whisper encoder) to the persistent kernel, whose LDS-DMA uses buffer
    descriptors: operands that are column slices of wider buffers (ld > K), M / N that are not tile multiples (rows past the
    end must read as zeros through the descriptor's range check, never as the NaNs placed right behind the views), fused
    epilogues, several tiles per workgroup."""
    ops = _ops()
    g = torch.Generator(device=dev).manual_seed(5)
    a_full = torch.full((M + 300, K + 192), float("nan"), device=dev, dtype=torch.bfloat16)
    b_full = torch.full((N + 300, K + 64), float("nan"), device=dev, dtype=torch.bfloat16)
    a_full[:M, 64:64 + K] = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    b_full[:N, :K] = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.bfloat16)
    a, b = a_full[:M, 64:64 + K], b_full[:N, :K]
    assert ops.gemm_kernel_name(M, N, K).startswith("gemm_nt_persist2") or M * N < 128 * 128 * 600
    bias = torch.randn(N, generator=g, device=dev)
    res = torch.randn(M, N, generator=g, device=dev).to(torch.bfloat16)
    ref = a.float() @ b.float().T
    for cfg in (0, 7):
        ops.gemm_set_config(cfg)
        try:
            c = ops.gemm_nt(a, b)
            assert_close(c, ref, atol=2e-2 * math.sqrt(K / 64), rtol=1e-2, what=f"cfg{cfg} plain")
            c = ops.gemm_nt(a, b, bias=bias, act=ops.ACT_GELU, residual=res)
            want = torch.nn.functional.gelu(ref + bias) + res.float()
            assert_close(c, want, atol=3e-2 * math.sqrt(K / 64), rtol=1e-2, what=f"cfg{cfg} bias+gelu+residual")
            acc = torch.ones((M, N), device=dev, dtype=torch.float32)
            ops.gemm_nt(a, b, out=acc, accumulate=True, alpha=0.5)
            assert_close(acc, 1.0 + 0.5 * ref, atol=1e-3 * math.sqrt(K / 64), rtol=1e-3, what=f"cfg{cfg} fp32 accumulate")
        finally:
            ops.gemm_set_config(0)


@pytest.mark.parametrize("M,N,K", [(516, 600, 448), (260, 256, 4096), (2 * 256 + 16, 1000, 192), (257, 260, 128)])
def test_gemm_thin_tail_tile_of_the_4wave_kernel(dev, M, N, K):
    """the 4-wave kernel computes an M-tail tile of at most 16 rows (31 x 380 tokens leave 4) straight from global memory
    (no staging): same results as the full-tile path with every epilogue, NaNs behind the views never read"""
    ops = _ops()
    g = torch.Generator(device=dev).manual_seed(9)
    a_full = torch.full((M + 40, K + 64), float("nan"), device=dev, dtype=torch.bfloat16)
    b_full = torch.full((N + 40, K + 128), float("nan"), device=dev, dtype=torch.bfloat16)
    a_full[:M, :K] = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    b_full[:N, 64:64 + K] = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.bfloat16)
    a, b = a_full[:M, :K], b_full[:N, 64:64 + K]
    bias = torch.randn(N, generator=g, device=dev)
    res = torch.randn(M, N, generator=g, device=dev).to(torch.bfloat16)
    ref = a.float() @ b.float().T
    ops.gemm_set_config(12)
    try:
        assert_close(ops.gemm_nt(a, b), ref, atol=2e-2 * math.sqrt(K / 64), rtol=1e-2, what="plain")
        c = ops.gemm_nt(a, b, bias=bias, act=ops.ACT_GELU, residual=res, alpha=0.5)
        assert_close(c, torch.nn.functional.gelu(0.5 * ref + bias) + res.float(), atol=3e-2 * math.sqrt(K / 64), rtol=1e-2, what="bias+gelu+res")
        acc = torch.ones((M, N + 8), device=dev, dtype=torch.float32)
        ops.gemm_nt(a, b, out=acc[:, :N], accumulate=True, alpha=0.5)
        assert_close(acc[:, :N], 1.0 + 0.5 * ref, atol=1e-3 * math.sqrt(K / 64), rtol=1e-3, what="fp32 accumulate")
        assert torch.all(acc[:, N:] == 1.0)
    finally:
        ops.gemm_set_config(0)


@pytest.mark.parametrize("where", ["below", "beyond"])
def test_gemm_operand_at_the_4gib_descriptor_boundary(dev, where):
    """the descriptor (LDS-DMA) GEMM forms address an operand with 32-bit byte offsets; a 288 GB-sized ragged
    batch puts M x ld near that limit (65 k tokens x the 28 672-wide gate|up stash = 3.7 GB).  `below`: the A operand ends a few rows
    under 4 GiB and is followed DIRECTLY by NaNs -- the descriptor forms run (cfg 12 and cfg 7), rows past M must read as zeros through
    the range check; `beyond`: the operand crosses 4 GiB -- `fits_descriptor` must send cfg 12 / cfg 7 / the auto rule to the kernels
    that address with 64-bit pointers, and the rows behind the boundary must be right (a wrapped 32-bit offset would read row 0's
    neighbourhood instead).  Row samples at the start, around the boundary and at the end against fp32 torch."""
    ops = _ops()
    K, N, ld = 14336, 384, 14400            # Llama-3-8B's h_mid with its LoRA extension columns: ld = 14336 + 64
    rows_4g = (1 << 32) // (2 * ld)         # 149 130 rows reach 4 GiB
    M = rows_4g - 300 if where == "below" else rows_4g + 2100
    g = torch.Generator(device=dev).manual_seed(11)
    buf = torch.full(((M + 64) * ld,), float("nan"), device=dev, dtype=torch.bfloat16)
    a = buf[: M * ld].view(M, ld)[:, :K]
    for r0 in range(0, M, 16384):           # (filled in slabs: a [M, K] fp32 temporary would be 8.5 GB)
        r1 = min(M, r0 + 16384)
        a[r0:r1] = torch.randn(r1 - r0, K, generator=g, device=dev).to(torch.bfloat16)
    buf[: M * ld].view(M, ld)[:, K:] = 0
    b = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.bfloat16)
    sample = torch.cat([torch.arange(0, 300), torch.arange(rows_4g - 600, min(M, rows_4g + 300)), torch.arange(M - 300, M)]).unique().to(dev)
    ref = a[sample].float() @ b.float().T
    for cfg in (0, 12, 7, 6):
        ops.gemm_set_config(cfg)
        try:
            c = ops.gemm_nt(a, b)
        finally:
            ops.gemm_set_config(0)
        assert torch.isfinite(c).all(), f"cfg {cfg}: a NaN from behind the operand reached the output"
        assert_close(c[sample], ref, atol=2e-2 * math.sqrt(K / 64), rtol=1e-2, what=f"cfg {cfg}, operand {where} 4 GiB")
    # the dX-shaped product of the same batch: the wide operand is the OUTPUT / residual side ([M, 2F] = 28 672 columns: 65 k rows = 3.7 GB)
    del buf, a, c


def test_gemm_asymmetric_identity(dev):
    """A = I with an asymmetric B catches row/col swaps in the MFMA output mapping."""
    ops = _ops()
    K = 128
    a = torch.eye(K, dtype=torch.bfloat16, device=dev)
    b = (torch.arange(192 * K, dtype=torch.float32).reshape(192, K) % 251 - 125).to(torch.bfloat16).to(dev)
    c = ops.gemm_nt(a, b, out_dtype=torch.float32)
    assert torch.equal(c.cpu(), b.float().T.cpu())


@pytest.mark.parametrize("act", [0, 1, 2])
def test_gemm_epilogues(dev, act):
    ops = _ops()
    M, N, K = 333, 260, 128
    a, b = rnd((M, K), dev, seed=3), rnd((N, K), dev, seed=4, std=0.1)
    bias = rnd((N,), dev, seed=5, dtype=torch.float32)
    res = rnd((50, N), dev, seed=6)
    c = ops.gemm_nt(a, b, bias=bias, residual=res, res_row_mod=50, act=act, alpha=0.5)
    ref = 0.5 * (a.float() @ b.float().T) + bias
    ref = F.gelu(ref) if act == 1 else (F.relu(ref) if act == 2 else ref)
    ref = ref + res.float()[torch.arange(M) % 50]
    assert_close(c, ref, atol=2e-2, rtol=1e-2, what=f"epilogue act{act}")
    # fp32 output + accumulate, strided operands (lda/ldb/ldc > logical width)
    big_a = rnd((M, K + 64), dev, seed=7)
    big_c = torch.ones((M, N + 8), dtype=torch.float32, device=dev)
    ops.gemm_nt(big_a[:, :K], b, out=big_c[:, :N], accumulate=True)
    ref2 = 1.0 + big_a[:, :K].float() @ b.float().T
    assert_close(big_c[:, :N], ref2, atol=1e-3, rtol=1e-4, what="f32 accumulate")
    assert torch.all(big_c[:, N:] == 1.0)


@pytest.mark.parametrize("M,F_,K", [(300, 192, 128), (1000, 1088, 256), (40, 64, 64)])
def test_gemm_swiglu_backward_epilogue_equals_separate_pass(dev, M, F_, K):
    """act = 3: dL/dh = dy . W never leaves the GEMM; the epilogue reproduces gemm -> swiglu_bwd (same formula on the
    bf16-rounded product; the two translation units may contract a*b+c differently: <= 1 bf16 ulp, almost all equal)"""
    ops = _ops()
    dy, wt = rnd((M, K), dev, seed=41), rnd((F_, K), dev, seed=42, std=K ** -0.5)
    gu = rnd((M, 2 * F_), dev, seed=43)
    want = ops.swiglu_bwd(gu, ops.gemm_nt(dy, wt))
    got = torch.zeros((M, 2 * F_), dtype=torch.bfloat16, device=dev)
    ops.gemm_nt(dy, wt, out=got[:, :F_], act=ops.ACT_SWIGLU_BWD, residual=gu)
    assert_close(got, want, atol=1e-6, rtol=2 ** -7, what="fused swiglu backward")
    assert float((got != want).float().mean()) < 0.01


@pytest.mark.parametrize("M,F_,K,forced", [(772, 1024, 4096, True), (600, 1088, 2112, True), (11780, 14336, 4096, False)])
def test_gemm_swiglu_forward_epilogue_equals_separate_pass(dev, M, F_, K, forced):
    """act 4 (slam_gemm_swiglu_bf16_nt): ONE launch writes the [gate64 | up64]-block stash and h = silu(gate) * up.  Against
    gemm_nt (same 4-wave kernel, plain [gate | up] weight) -> swiglu_fwd: the product's elements are the same dot products in the
    same k order and silu runs on the bf16-rounded values in both, so BOTH outputs must be bit-identical once the column blocks are
    put back in order.  Shapes: thin M tail (772 = 3 x 256 + 4), a ragged last tile row (600) with a ragged last tile column
    (N = 2176 = 8.5 x 256), an odd number of k-tiles (K = 33 x 64); the Llama-3-8B shape of the C3 batch under the AUTO rule.  Then the backward
    elementwise pass reading the block layout == reading the plain one."""
    ops = _ops()
    x = rnd((M, K), dev, seed=51)
    w = rnd((2 * F_, K), dev, seed=52, std=K ** -0.5)
    try:
        if forced:
            ops.gemm_set_config(12)
        assert ops.gemm_swiglu_supported(M, 2 * F_, K, K, K)
        gu_ref = ops.gemm_nt(x, w)
        assert "w4" in ops.gemm_kernel_name(M, 2 * F_, K)
        h_ref = ops.swiglu_fwd(gu_ref)
        wil = ops.interleave_gate_up(w)
        gu_il = torch.full((M, 2 * F_), float("nan"), dtype=torch.bfloat16, device=dev)
        h = torch.full((M, F_), float("nan"), dtype=torch.bfloat16, device=dev)
        ops.gemm_swiglu(x, wil, gu_il, h)
    finally:
        ops.gemm_set_config(0)
    blocks = gu_il.view(M, F_ // 64, 2, 64)
    gu_back = torch.cat([blocks[:, :, 0].reshape(M, F_), blocks[:, :, 1].reshape(M, F_)], dim=1)
    assert torch.equal(gu_back, gu_ref), f"stash differs in {int((gu_back != gu_ref).sum())} elements"
    assert torch.equal(h, h_ref), f"h differs in {int((h != h_ref).sum())} elements"
    dh = rnd((M, F_), dev, seed=53)
    assert torch.equal(ops.swiglu_bwd(gu_il, dh, interleaved=True), ops.swiglu_bwd(gu_ref, dh))


def test_gemm_swiglu_forward_is_refused_where_the_4_wave_kernel_does_not_run(dev):
    ops = _ops()
    from slam_llm_amd.lib import SlamHipError
    assert not ops.gemm_swiglu_supported(11780, 11264, 2048, 2048, 2048)      # K <= 2048: the persistent kernel's territory
    assert not ops.gemm_swiglu_supported(64, 28672, 4096, 4096, 4096)         # small M: 128x128 tiles win the auto rule
    x, w = rnd((64, 4096), dev, seed=1), rnd((2048, 4096), dev, seed=2)
    with pytest.raises(SlamHipError, match="not served"):
        ops.gemm_swiglu(x, w, torch.empty((64, 2048), dtype=torch.bfloat16, device=dev), torch.empty((64, 1024), dtype=torch.bfloat16, device=dev))


@pytest.mark.parametrize("M,N,K,slices,epi", [
    (4352, 4096, 4096, 0, "bias+res"),      # 17 x 16 = 272 tiles = one round + 16: the AUTO plan (8 slices of 8 k-tiles)
    (4352, 4096, 4096, 3, "none"),          # forced 3 slices of 21 / 21 / 22 k-tiles
    (600, 1024, 2112, 3, "gelu"),           # under-filled grid (12 tiles, ragged last tile row), 33 k-tiles in 3 slices of 11
    (772, 512, 4096, 4, "f32acc"),          # thin 4-row tail tiles among the split tiles; fp32 accumulate epilogue
    (2048, 2048, 4096, 2, "none"),          # exactly 64 tiles, every one split in two
])
@pytest.mark.parametrize("form", ["in-launch", "two-launch"])
def test_gemm_split_k_tail(dev, M, N, K, slices, epi, form):
    """split-K of the 4-wave kernel -- in-launch form (partial slabs + ticket + fixed-order fix-up by the last arriver; round 3) and
    two-launch form (the slices only store their slabs, gemm_sk_reduce_kernel adds them in slice order and runs the epilogue; round 4,
    what the auto rule picks for mid-M products) -- against the unsplit launch of the same kernel (fp32 partial sums are re-associated:
    bf16 outputs may differ by one rounding on a few elements: <= 2^-7 relative, < 2 % of the elements), against an fp32 matmul, and
    bit-reproducible from run to run."""
    ops = _ops()
    a, b = rnd((M, K), dev, seed=61), rnd((N, K), dev, seed=62, std=K ** -0.5)
    bias = torch.randn(N, device=dev) if epi in ("bias+res", "gelu") else None
    res = rnd((M, N), dev, seed=63) if epi == "bias+res" else None
    act = ops.ACT_GELU if epi == "gelu" else ops.ACT_NONE
    f32 = epi == "f32acc"
    base = torch.randn(M, N, device=dev) if f32 else None

    def run(mode):
        ops.gemm_set_config(mode)
        out = base.clone() if f32 else torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
        ops.gemm_nt(a, b, out=out, bias=bias, residual=res, act=act, accumulate=f32)
        return out
    if form == "two-launch" and slices == 0:
        pytest.skip("the two-launch form has no tail auto plan: its auto rule covers under-filled grids (test_gemm_mid_m_rule_*)")
    try:
        ops.gemm_set_config(12)
        ops.gemm_set_config(362 if form == "two-launch" else 361)
        if slices == 0:   # the shipped auto plan is conservative (tail <= 32 tiles, 2 slices); exercise the general planner too
            ops.gemm_set_config(320 + 16)   # tails up to 128 tiles
            ops.gemm_set_config(340 + 8)    # up to 8 slices
        plain = run(301)
        split1 = run(300 + slices if slices else 300)
        split2 = run(300 + slices if slices else 300)
    finally:
        ops.gemm_set_config(320 + 4)
        ops.gemm_set_config(340 + 2)
        ops.gemm_set_config(301)    # shipped default: split-K tail off
        ops.gemm_set_config(361)    # shipped default: two-launch form by the auto rule only
        ops.gemm_set_config(0)
    assert torch.equal(split1, split2), "split-K tail is not reproducible"
    assert bool(torch.isfinite(split1).all())
    ref = a.float() @ b.float().t()
    if bias is not None:
        ref = ref + bias
    if act == ops.ACT_GELU:
        ref = F.gelu(ref)
    if res is not None:
        ref = ref + res.float()
    if f32:
        ref = ref + base
    assert_close(split1, ref, atol=2e-2, rtol=2e-2, what="split-K vs fp32")
    # (re-associated fp32 partial sums: one bf16 rounding step of the OUTPUT's magnitude, or -- where bias / residual cancel the
    # product -- of the operands' magnitude, O(1..8) here: 2^-5)
    assert_close(split1, plain, atol=1e-3 if f32 else 2 ** -5, rtol=2 ** -7, what="split-K vs unsplit")
    if not f32:
        assert float((split1 != plain).float().mean()) < 0.02
    if slices == 0:   # the auto plan must actually have split this shape: 272 tiles leave 16 for the last round
        assert (M, N) == (4352, 4096)


@pytest.mark.parametrize("M,N,K,epi", [(672, 4096, 11008, "res"), (672, 4096, 4096, "none"), (380, 2048, 5632, "bias+res"), (1520, 4096, 14336, "f32acc"),
                                       (100, 1024, 8192, "gelu")])
def test_gemm_mid_m_rule_slices_k_and_matches_unsliced(dev, M, N, K, epi):
    """AUTO rule, mid-M products (C4's M = 672, C1's M = 380, half a C2 batch: fewer 256 x 256 tiles than half the CUs): the rule picks
    the 4-wave kernel on K slices + the reduce launch (ops.gemm_kernel_name says so); result vs the same product with the rule's
    K-slicing off (one rounding of re-associated fp32 sums), vs fp32, and bit-reproducible."""
    ops = _ops()
    a, b = rnd((M, K), dev, seed=81), rnd((N, K), dev, seed=82, std=K ** -0.5)
    bias = torch.randn(N, device=dev) if epi in ("bias+res", "gelu") else None
    res = rnd((M, N), dev, seed=83) if epi in ("bias+res", "res") else None
    act = ops.ACT_GELU if epi == "gelu" else ops.ACT_NONE
    f32 = epi == "f32acc"
    base = torch.randn(M, N, device=dev) if f32 else None

    def run():
        out = base.clone() if f32 else torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
        ops.gemm_nt(a, b, out=out, bias=bias, residual=res, act=act, accumulate=f32)
        return out
    assert ops._GEMM_CFG == 0
    run()      # (registers the scratch buffer on the first product of the process)
    assert "split-K" in ops.gemm_kernel_name(M, N, K), ops.gemm_kernel_name(M, N, K)
    s1, s2 = run(), run()
    try:
        ops.gemm_set_config(360)
        assert "split-K" not in ops.gemm_kernel_name(M, N, K)
        plain = run()
    finally:
        ops.gemm_set_config(361)
    assert torch.equal(s1, s2) and bool(torch.isfinite(s1).all())
    ref = a.float() @ b.float().t()
    if bias is not None:
        ref = ref + bias
    if act == ops.ACT_GELU:
        ref = F.gelu(ref)
    if res is not None:
        ref = ref + res.float()
    if f32:
        ref = ref + base
    assert_close(s1, ref, atol=2e-2, rtol=2e-2, what="K-sliced vs fp32")
    assert_close(s1, plain, atol=1e-3 if f32 else 2 ** -5, rtol=2 ** -7, what="K-sliced vs unsliced")


@pytest.mark.parametrize("M,N,K,mode", [(672, 64, 12288, "bf16"), (380, 16, 2560, "bf16"), (3040, 64, 6144, "f32acc"), (70, 32, 512, "bias"), (4096, 64, 4096, "bf16")])
def test_gemm_tall_skinny_k_sliced(dev, M, N, K, mode):
    """N <= 64, M <= 4096 (the LoRA-extension columns of the dX products at the small-batch recipes, first hops with r not a multiple of
    4): 64-row x K-slice workgroups + fixed-order reduce, vs the 128 x 64 tile kernel (gemm_set_config 370) and fp32; bit-reproducible."""
    ops = _ops()
    a, b = rnd((M, K), dev, seed=91), rnd((N, K), dev, seed=92, std=K ** -0.5)
    bias = torch.randn(N, device=dev) if mode == "bias" else None
    f32 = mode == "f32acc"
    base = torch.randn(M, N, device=dev) if f32 else None

    def run():
        out = base.clone() if f32 else torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
        ops.gemm_nt(a, b, out=out, bias=bias, accumulate=f32, alpha=0.5 if f32 else 1.0)
        return out
    assert ops.gemm_kernel_name(M, N, K) == "gemm_ts_kernel"
    t1, t2 = run(), run()
    try:
        ops.gemm_set_config(370)
        tile = run()
    finally:
        ops.gemm_set_config(371)
    assert torch.equal(t1, t2) and bool(torch.isfinite(t1).all())
    ref = (0.5 if f32 else 1.0) * (a.float() @ b.float().t())
    if bias is not None:
        ref = ref + bias
    if f32:
        ref = ref + base
    assert_close(t1, ref, atol=2e-2, rtol=2e-2, what="tall-skinny vs fp32")
    assert_close(t1, tile, atol=1e-3 if f32 else 2 ** -5, rtol=2 ** -7, what="tall-skinny vs tile kernel")


def test_gemm_swiglu_forward_with_split_k_tail(dev):
    """the fused SwiGLU-forward epilogue behind the split-K fix-up (the last arriver of a tile runs it on the summed slabs)"""
    ops = _ops()
    M, F_, K = 1100, 1024, 4096          # 5 x 8 = 40 tiles, thin? no: 1100 = 4 x 256 + 76
    x, w = rnd((M, K), dev, seed=71), rnd((2 * F_, K), dev, seed=72, std=K ** -0.5)
    wil = ops.interleave_gate_up(w)
    outs = []
    try:
        ops.gemm_set_config(12)
        for mode in (301, 304):
            ops.gemm_set_config(mode)
            gu = torch.full((M, 2 * F_), float("nan"), dtype=torch.bfloat16, device=dev)
            h = torch.full((M, F_), float("nan"), dtype=torch.bfloat16, device=dev)
            ops.gemm_swiglu(x, wil, gu, h)
            outs.append((gu, h))
    finally:
        ops.gemm_set_config(301)
        ops.gemm_set_config(0)
    assert_close(outs[1][0], outs[0][0], atol=1e-6, rtol=2 ** -7, what="stash, split vs unsplit")
    assert_close(outs[1][1], outs[0][1], atol=1e-3, rtol=2 ** -6, what="h, split vs unsplit")
    g, u = x.float() @ w[:F_].float().t(), x.float() @ w[F_:].float().t()
    assert_close(outs[1][1], F.silu(g) * u, atol=2e-2, rtol=3e-2, what="h vs fp32")


def test_gemm_rejects_bad_shapes(dev):
    ops = _ops()
    from slam_llm_amd.lib import SlamHipError
    with pytest.raises(SlamHipError, match="64"):
        ops.gemm_nt(rnd((8, 40), dev), rnd((8, 40), dev))
    with pytest.raises(SlamHipError, match="HBM"):
        ops.gemm_nt(torch.zeros(8, 64, dtype=torch.bfloat16), torch.zeros(8, 64, dtype=torch.bfloat16))


# ----------------------------------------------------------------------------------------- attention: relpos-bias backward
@pytest.mark.parametrize("T,masked,drop_p", [(150, False, 0.0), (200, True, 0.0), (150, False, 0.1), (200, True, 0.1)])
def test_attn_bwd_with_gated_relative_position_bias(dev, T, masked, drop_p):
    """unfrozen WavLM (modules.py:504-533): score = scale q.k + gate[b,h,q] * table[h][k - q + T - 1].  The backward kernels recompute P with
    the bias; the dQ kernel materialises dL/d(score), from which d(gate) (row sums against the table) and d(table) (diagonal sums weighted by
    the gate, accumulated) are reduced.  Against torch autograd in fp32: dQ / dK / dV cosine >= 0.999 and max err <= 3e-2 max|ref|; d(gate),
    d(table) cosine >= 0.999 (d(table) checked after TWO backward calls = twice the single-call value: it accumulates).
    drop_p > 0: attention_dropout of the un-frozen WavLM in train mode on top of the bias (the RP + DROP instantiations of the three
    kernels); the reference gets the kernels' mask rebuilt on the host."""
    ops = _ops()
    B, H, D = 2, 3, 64
    Tp = ops.round_up(T, 64)
    g = torch.Generator().manual_seed(5)
    qkv = (torch.randn(B * T, 3 * H * D, generator=g) * 1.0).to(torch.bfloat16).to(dev)
    q2d, k2d, v2d = qkv[:, : H * D], qkv[:, H * D: 2 * H * D], qkv[:, 2 * H * D:]
    do2d = torch.randn(B * T, H * D, generator=g).to(torch.bfloat16).to(dev)
    gate = (1.0 + torch.rand(B, H, T, generator=g)).to(dev)
    tabv = (torch.randn(H, 2 * T - 1, generator=g) * 0.5).to(dev)
    km = None
    if masked:
        km = torch.zeros((B, Tp), dtype=torch.uint8, device=dev)
        km[0, :T] = 1
        km[1, : T - 37] = 1
    scale = D ** -0.5

    def tr(x2d):
        t = torch.zeros((B, H, D, Tp), dtype=torch.bfloat16, device=dev)
        t[..., :T] = x2d.view(B, T, H, D).permute(0, 2, 3, 1)
        return t
    gate_p = torch.zeros((B, H, Tp), dtype=torch.float32, device=dev)
    gate_p[..., :T] = gate
    tab = ops.relpos_table(tabv)
    vt, qt, kt, dot = tr(v2d), tr(q2d), tr(k2d), tr(do2d)
    drop = (drop_p, 0xABCDE12345) if drop_p > 0 else None
    o2d, lse = ops.attn_fwd(q2d, k2d, vt, B, T, H, H, D, False, scale, key_mask=km, relpos=(gate_p, tab, T), drop=drop)
    dq, dk, dv = torch.empty_like(q2d), torch.empty_like(k2d), torch.empty_like(v2d)
    d_gate = torch.zeros_like(gate_p)
    d_tab = torch.zeros_like(tab)
    for _ in range(2):
        ops.attn_bwd(q2d, k2d, v2d, o2d, do2d, lse, dq, dk, dv, B, T, H, H, D, False, scale, key_mask=km,
                     relpos=(gate_p, tab, T, d_gate, d_tab), drop=drop)
    # fp32 reference
    qf = q2d.float().view(B, T, H, D).transpose(1, 2).detach().requires_grad_(True)
    kf = k2d.float().view(B, T, H, D).transpose(1, 2).detach().requires_grad_(True)
    vf = v2d.float().view(B, T, H, D).transpose(1, 2).detach().requires_grad_(True)
    gr = gate.clone().requires_grad_(True)
    tr_ = tabv.clone().requires_grad_(True)
    idx = (torch.arange(T, device=dev)[None, :] - torch.arange(T, device=dev)[:, None] + T - 1)      # [q, k] -> k - q + T - 1
    sc = (qf @ kf.transpose(2, 3)) * scale + gr[..., None] * tr_[:, idx][None]
    if masked:
        sc = sc.masked_fill(km[:, None, None, :T] == 0, float("-inf"))
    pr = torch.softmax(sc, -1)
    if drop is not None:
        keep = torch.from_numpy(G.attn_keep_mask(drop[1], drop_p, B, H, T, T, Tp, Tp)).to(dev)
        assert abs(float(keep.mean()) - (1 - drop_p)) < 0.01
        pr = pr * keep / (1 - drop_p)
    o_ref = pr @ vf
    (o_ref * do2d.float().view(B, T, H, D).transpose(1, 2)).sum().backward()
    back = lambda t: t.transpose(1, 2).reshape(B * T, H * D)  # noqa: E731
    assert_close(o2d, back(o_ref.detach()), atol=2e-2, rtol=2e-2, what="forward with bias")
    for name, got, ref in (("dQ", dq, back(qf.grad)), ("dK", dk, back(kf.grad)), ("dV", dv, back(vf.grad))):
        cs = float((got.float() * ref).sum() / (got.float().norm() * ref.norm() + 1e-30))
        err = float((got.float() - ref).abs().max())
        G.floor_check(cs, 0.999, f"{name}: cosine {cs}, max err {err}")
        assert err <= 3e-2 * float(ref.abs().max()), f"{name}: cosine {cs}, max err {err}"
    cs = float((d_gate[..., :T] * gr.grad).sum() / (d_gate[..., :T].norm() * gr.grad.norm() + 1e-30))
    G.floor_check(cs, 0.999, f"d(gate): cosine {cs}")
    assert abs(float(d_gate[..., :T].norm()) / float(gr.grad.norm()) - 1) < 3e-2, f"d(gate): cosine {cs}"
    got_t = d_tab[:, 64: 64 + 2 * T - 1]
    cs = float((got_t * 2 * tr_.grad).sum() / (got_t.norm() * (2 * tr_.grad).norm() + 1e-30))
    G.floor_check(cs, 0.999, f"d(table): cosine {cs}")
    assert abs(float(got_t.norm()) / float((2 * tr_.grad).norm()) - 1) < 3e-2, f"d(table): cosine {cs}"
    assert float(d_tab[:, :64].abs().max()) == 0 and float(d_tab[:, 64 + 2 * T - 1:].abs().max()) == 0      # the slack stays untouched


@pytest.mark.parametrize("cfg", [0, 1, 2, 6, 7, 11, 12])
@pytest.mark.parametrize("k,stride,C,T", [(3, 2, 512, 4001), (2, 2, 512, 1500), (3, 2, 64, 777)])
def test_gemm_over_the_overlapping_row_window_view_equals_im2col(dev, cfg, k, stride, C, T):
    """lda < K (include/slam_hip.h): A = the window view of a row-major [T, C] signal (row t = the k * C contiguous elements from row
    stride * t on) -- the product must equal, bit for bit, the product over the materialised im2col matrix with the same kernel, for
    every GEMM kernel the auto rule can pick (HuBERT / WavLM conv layers 1-6 run this way), incl. bias + GELU and a row count that is
    not a tile multiple; and it must equal torch's conv1d to bf16 rounding."""
    ops = _ops()
    g = torch.Generator().manual_seed(k * 100 + C)
    x = torch.randn(T + 4, C, generator=g).to(torch.bfloat16).to(dev)        # (+ slack rows: the view never reads past row T - 1)
    Wt = (torch.randn(256, C, k, generator=g) * (C * k) ** -0.5)
    w2 = Wt.permute(0, 2, 1).reshape(256, k * C).to(torch.bfloat16).to(dev)     # tap-major columns = the window's element order
    bias = (torch.randn(256, generator=g) * 0.1).to(dev)
    Tout = (T - k) // stride + 1
    a_view = x.as_strided((Tout, k * C), (stride * C, 1), x.storage_offset())
    cols, To2 = ops.conv1d_im2col(x[:T], 1, T, 0, C, k, stride, 0, Kp=k * C)
    assert To2 == Tout and torch.equal(cols, a_view.contiguous())
    ops.gemm_set_config(cfg)
    try:
        y_view = ops.gemm_nt(a_view, w2, bias=bias, act=ops.ACT_GELU)
        y_cols = ops.gemm_nt(cols, w2, bias=bias, act=ops.ACT_GELU)
    finally:
        ops.gemm_set_config(0)
    assert torch.equal(y_view, y_cols)
    ref = F.gelu(F.conv1d(x[:T].float().t()[None].cpu(), Wt.to(torch.bfloat16).float(), bias.cpu(), stride=stride))[0].t()
    assert_close(y_view, ref, atol=2e-2, rtol=2e-2, what="window-view conv vs torch")


# ----------------------------------------------------------------------------------------- grouped positional conv
@pytest.mark.parametrize("C,K,T", [(64, 128, 300), (48, 128, 517), (32, 16, 100), (64, 127, 256), (80, 128, 300)])
def test_pos_conv_one_launch_matches_torch_grouped_conv(dev, C, K, T):
    """slam_pos_conv_fwd (implicit GEMM: taps = LDS row offsets) against torch's grouped Conv1d + SamePad + GELU + residual in fp32
    (fairseq pos_conv, WavLM.py:378-386 / 575-580): channels per group 64 / 48 / 32 / 80 (d = 1024 / 768 / 512 / 1280 with 16 groups), even and
    odd tap counts, T across tile boundaries; |err| <= 2e-2 + 2e-2 |ref| (bf16 output).  Also against the round-2 path
    (per-group im2col + GEMM), which must agree to bf16 rounding."""
    ops = _ops()
    B, G = 2, 4
    d = G * C
    h = rnd((B * T, d), dev, seed=81)
    Wt = (torch.randn(d, C, K, generator=torch.Generator().manual_seed(82)) * (C * K) ** -0.5)
    bias = torch.randn(d, generator=torch.Generator().manual_seed(83)) * 0.1
    w_im2col = Wt.view(G, C, C, K).permute(0, 1, 3, 2).reshape(G, C, K * C).to(torch.bfloat16).to(dev)
    assert ops.pos_conv_supported(C, K)
    x = torch.full((B * T, d), float("nan"), dtype=torch.bfloat16, device=dev)
    ops.pos_conv_fwd(h, ops.pos_conv_pack(w_im2col, K), bias.to(dev), B, T, out=x)
    hf = h.float().cpu().view(B, T, d)
    conv = F.conv1d(hf.transpose(1, 2), Wt.to(torch.bfloat16).float(), bias, padding=K // 2, groups=G)[..., :T]   # SamePad: drop the extra frame of even kernels
    ref = hf + F.gelu(conv.transpose(1, 2))
    assert_close(x.view(B, T, d), ref, atol=2e-2, rtol=2e-2, what="pos_conv vs torch")
    # the round-2 path: per group im2col + GEMM with the GELU / residual epilogue
    Kp = ops.round_up(K * C, 64)
    wpad = torch.zeros((G, C, Kp), dtype=torch.bfloat16, device=dev)
    wpad[:, :, : K * C] = w_im2col
    x2 = torch.empty_like(x)
    cols = torch.empty((B * T, Kp), dtype=torch.bfloat16, device=dev)
    for g in range(G):
        ops.conv1d_im2col(h, B, T, g * C, C, K, 1, K // 2, Kp=Kp, Tout_limit=T, out=cols)
        ops.gemm_nt(cols, wpad[g], out=x2[:, g * C:(g + 1) * C], bias=bias.to(dev)[g * C:(g + 1) * C].contiguous(), act=ops.ACT_GELU,
                    residual=h[:, g * C:(g + 1) * C])
    assert_close(x, x2, atol=2e-2, rtol=2 ** -6, what="pos_conv vs im2col + GEMM")


# ----------------------------------------------------------------------------------------- norms
@pytest.mark.parametrize("M,d", [(37, 64), (1000, 1280), (5, 4096)])
def test_layernorm(dev, M, d):
    ops = _ops()
    x = rnd((M, d), dev, seed=1, std=2.0)
    w, b = rnd((d,), dev, seed=2, dtype=torch.float32), rnd((d,), dev, seed=3, dtype=torch.float32)
    y = ops.layernorm(x, w, b, 1e-5)
    ref = F.layer_norm(x.float(), (d,), w, b, 1e-5)
    assert_close(y, ref, atol=1e-2, rtol=1e-2, what="layernorm")


@pytest.mark.parametrize("M,d,gelu", [(4101, 1280, False), (4099, 512, True), (8191, 1024, False)])
def test_layernorm_two_rows_per_wave_is_bit_identical(dev, M, d, gelu):
    """round 6: batches of >= 4096 rows of d <= 1536 (the Whisper / HuBERT / WavLM encoders) run the two-rows-per-wave kernel; the same rows through the one-row
    kernel (in pieces of < 4096 rows) give the same bits for y, mean and rstd -- odd row counts and a strided input included -- and both meet torch"""
    ops = _ops()
    xs = rnd((M, d + 8), dev, seed=11, std=2.0)
    x = xs[:, :d]                                    # (row pitch d + 8: not contiguous)
    w, b = rnd((d,), dev, seed=12, dtype=torch.float32), rnd((d,), dev, seed=13, dtype=torch.float32)
    y = torch.full((M, d), float("nan"), dtype=torch.bfloat16, device=dev)
    y, mean, rstd = ops.layernorm(x, w, b, 1e-5, out=y, gelu=gelu, stats=True)
    ys, ms, rs = [], [], []
    for r0 in range(0, M, 3000):
        yy, mm, rr = ops.layernorm(x[r0:r0 + 3000], w, b, 1e-5, gelu=gelu, stats=True)
        ys.append(yy), ms.append(mm), rs.append(rr)
    assert torch.equal(y.view(torch.int16), torch.cat(ys).view(torch.int16))
    assert torch.equal(mean, torch.cat(ms)) and torch.equal(rstd, torch.cat(rs))
    ref = F.layer_norm(x.float(), (d,), w, b, 1e-5)
    if gelu:
        ref = F.gelu(ref)
    assert_close(y, ref, atol=1e-2, rtol=1e-2, what="layernorm (two rows per wave)")


@pytest.mark.parametrize("M,d", [(37, 128), (300, 4096), (16, 4096)])
def test_rmsnorm_fwd_bwd(dev, M, d):
    ops = _ops()
    x = rnd((M, d), dev, seed=1, std=1.5)
    w = (1 + 0.1 * torch.randn(d)).to(dev)
    dy = rnd((M, d), dev, seed=2)
    dres = rnd((M, d), dev, seed=3)
    y, rstd = ops.rmsnorm_fwd(x, w, 1e-5)
    xf = x.float().requires_grad_(True)
    ref = w * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5))
    assert_close(y, ref.detach(), atol=1e-2, rtol=1.2e-2, what="rmsnorm fwd")
    ref.backward(dy.float())
    scale = torch.tensor([0.5], device=dev)
    dx = ops.rmsnorm_bwd(x, rstd, w, dy, dres=dres, grad_scale=scale)
    assert_close(dx, 0.5 * xf.grad + dres.float(), atol=2e-2, rtol=1.5e-2, what="rmsnorm bwd")


# ----------------------------------------------------------------------------------------- rope / transposes
def _rope_ref(x, theta):  # x [B,T,H,D]
    B, T, H, D = x.shape
    inv = 1.0 / (theta ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
    fr = torch.arange(T, dtype=torch.float32)[:, None] * inv[None]
    emb = torch.cat([fr, fr], -1)
    cos, sin = emb.cos()[None, :, None, :], emb.sin()[None, :, None, :]
    rot = torch.cat([-x[..., D // 2:], x[..., : D // 2]], -1)
    return x * cos + rot * sin


@pytest.mark.parametrize("D", [64, 128])
def test_head_rope_transpose(dev, D):
    ops = _ops()
    B, T, H = 2, 75, 3
    ld = H * D + 2 * D + 64
    buf = rnd((B * T, ld), dev, seed=1)
    orig = buf.clone()
    inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
    fr = torch.arange(T, dtype=torch.float32)[:, None] * inv[None]
    cos, sin = fr.cos().contiguous().to(dev), fr.sin().contiguous().to(dev)
    xt = ops.head_rope_transpose(buf, 64, B, T, H, D, cos=cos, sin=sin)
    x = orig[:, 64:64 + H * D].float().cpu().view(B, T, H, D)
    ref = _rope_ref(x, 10000.0)
    got = buf[:, 64:64 + H * D].float().cpu().view(B, T, H, D)
    assert_close(got, ref, atol=2e-2, rtol=1e-2, what="rope in place")
    assert torch.equal(buf[:, :64], orig[:, :64]) and torch.equal(buf[:, 64 + H * D:], orig[:, 64 + H * D:])
    Tp = xt.shape[-1]
    assert Tp == 128
    assert torch.equal(xt[..., :T].cpu(), buf[:, 64:64 + H * D].cpu().view(B, T, H, D).permute(0, 2, 3, 1))
    assert torch.all(xt[..., T:] == 0)
    # inverse rotation restores the input (RoPE backward = transposed rotation)
    ops.head_rope_transpose(buf, 64, B, T, H, D, cos=cos, sin=sin, inverse=True, want_t=False)
    assert_close(buf[:, 64:64 + H * D], orig[:, 64:64 + H * D], atol=3e-2, rtol=2e-2, what="rope inverse")


def test_transpose(dev):
    ops = _ops()
    x = rnd((333, 200), dev, seed=1)
    t = ops.transpose(x)
    assert t.shape == (200, 384)
    assert torch.equal(t[:, :333].cpu(), x.cpu().T)
    assert torch.all(t[:, 333:] == 0)


# ----------------------------------------------------------------------------------------- attention
def _attn_ref(q, k, v, causal, kmask, scale):
    """q [B,T,Hq,D], k/v [B,T,Hkv,D] fp32 -> o [B,T,Hq,D]; rows with no visible key give 0."""
    B, T, Hq, D = q.shape
    Hkv = k.shape[2]
    rep = Hq // Hkv
    qh = q.permute(0, 2, 1, 3)
    kh = k.permute(0, 2, 1, 3).repeat_interleave(rep, dim=1)
    vh = v.permute(0, 2, 1, 3).repeat_interleave(rep, dim=1)
    s = qh @ kh.transpose(2, 3) * scale
    allowed = torch.ones(B, 1, T, T, dtype=torch.bool, device=q.device)
    if causal:
        allowed = allowed & torch.tril(torch.ones(T, T, dtype=torch.bool, device=q.device))[None, None]
    if kmask is not None:
        allowed = allowed & kmask.bool()[:, None, None, :]
    s = s.masked_fill(~allowed, float("-inf"))
    p = torch.softmax(s, dim=-1)
    p = torch.nan_to_num(p, nan=0.0)
    return (p @ vh).permute(0, 2, 1, 3)


@pytest.fixture(params=["tr", "copies"])
def attn_form(request):
    """the attention tests below run twice: on the shipped kernels (transposed MFMA operands by ds_read_b64_tr_b16 from the row-major
    tiles: V is passed row-major, no [B,H,D,Tp] copy exists) and on the round-3 kernels (slam_attn_set_fwd_qf 40: V^T passed as a copy,
    ops.attn_bwd builds Q^T / K^T / dO^T)"""
    from slam_llm_amd.lib import call
    call("slam_attn_set_fwd_qf", 41 if request.param == "tr" else 40)
    yield request.param
    call("slam_attn_set_fwd_qf", 41)


def _prep_attn(ops, dev, B, T, Hq, Hkv, D, seed, form="tr"):
    """-> qkv, q2, k2, v2, Tp, v_arg: v_arg is what attn_fwd gets for V (row-major view | transposed copy)"""
    ld = (Hq + 2 * Hkv) * D
    qkv = rnd((B * T, ld), dev, seed=seed, std=1.0)
    q2, k2, v2 = qkv[:, : Hq * D], qkv[:, Hq * D:(Hq + Hkv) * D], qkv[:, (Hq + Hkv) * D:]
    vt = ops.head_rope_transpose(qkv, (Hq + Hkv) * D, B, T, Hkv, D) if form == "copies" else None
    return qkv, q2, k2, v2, ops.round_up(T, 64), (vt if form == "copies" else v2)


@pytest.mark.parametrize("B,T,Hq,Hkv,D,causal,masked", [
    (2, 150, 2, 2, 64, False, False),     # encoder-like, T not a multiple of the tile
    (1, 1500, 1, 1, 64, False, False),    # Whisper context length
    (2, 100, 4, 2, 64, True, True),       # causal GQA + left padding
    (2, 380, 4, 1, 128, True, True),      # Llama-3 head_dim, T of the C3 workload
    (1, 64, 2, 2, 128, True, False),
])
def test_attention_fwd(dev, attn_form, B, T, Hq, Hkv, D, causal, masked):
    ops = _ops()
    qkv, q2, k2, v2, Tp, vt = _prep_attn(ops, dev, B, T, Hq, Hkv, D, seed=11, form=attn_form)
    km = None
    km_ref = None
    if masked:
        km = torch.zeros((B, Tp), dtype=torch.uint8, device=dev)
        km[:, :T] = 1
        km[0, :7] = 0  # left padding on sample 0
        km_ref = km[:, :T]
    scale = D ** -0.5
    o, lse = ops.attn_fwd(q2, k2, vt, B, T, Hq, Hkv, D, causal, scale, key_mask=km)
    ref = _attn_ref(q2.float().view(B, T, Hq, D), k2.float().view(B, T, Hkv, D), v2.float().view(B, T, Hkv, D),
                    causal, km_ref, scale)
    # P is rounded to bf16 before the PV product and O to bf16: ~1e-2 relative on O(1) values
    assert_close(o.view(B, T, Hq, D), ref, atol=2e-2, rtol=2e-2, what="attn fwd")
    assert torch.isfinite(o.float()).all()


@pytest.mark.parametrize("B,T,H,grow", [
    (2, 150, 2, 0.0),      # T not a multiple of the tile: the keys past T of the last tile start their accumulators at -inf
    (1, 1500, 2, 0.0),     # Whisper context length (23 whole tiles + 28 keys)
    (3, 40, 2, 0.0),       # one tile that is both the first (ordinary path, zero start) and the partial last one
    (2, 64, 1, 0.0),       # exactly one whole tile
    (2, 128, 2, 0.0),      # whole tiles only
    (2, 700, 2, 13.0),     # spike keys whose scores exceed the running maximum by ~2^7, then ~2^14, ...: the attempt "P = exp2(S')"
    (1, 1500, 1, 13.0),    #   fails there and the scores are computed again (the fallback s_product); the maximum moves at every second spike
])
def test_attention_fwd_prescaled_q_form(dev, B, T, H, grow):
    """round 5: a frozen query projection may carry the softmax scale and the exponent's base change (W_q, b_q multiplied by
    scale * log2(e) in fp32 at load time: HipWhisperEncoder.load); attn_fwd(..., q_prescaled=True) -- a negative scale at the C ABI --
    then takes the scores as the first product delivers them, and LSE-less launches of the mask-free bidirectional D = 64 form run
    attn_fwd_kernel<..., QS>: accumulators of the first product started at -m (at -inf for the keys past T), P = exp2 of the product.
    Checked here with Q = bf16(c * q): vs the fp32 reference on the unscaled q at the tolerance of the other forward tests (widened by
    the test's own second rounding of Q where spikes make the scores large), vs the general softmax on the same pre-scaled Q (knob 60)
    tightly, and vs the LSE of the ordinary launch on the unscaled q (the LSE keeps its meaning)."""
    ops = _ops()
    from slam_llm_amd.lib import call
    D = 64
    qkv = rnd((B * T, 3 * H * D), dev, seed=31, std=1.0)
    if grow:
        # every query has a component 3 along a fixed unit direction; spike key i (t = 70 + 130 i) has a component grow * (i + 1) along
        # it: its score stands out by 3 * grow * (i + 1) / 8 * log2(e) = ~7 (i + 1) in the exponent's log2 units
        dirn = torch.nn.functional.normalize(torch.randn(D, generator=torch.Generator().manual_seed(5)), dim=0).to(dev)
        k3 = qkv[:, H * D: 2 * H * D].float().view(B, T, H, D).clone()
        for i, t in enumerate(range(70, T, 130)):
            k3[:, t] += grow * (i + 1) * dirn
        q3 = qkv[:, : H * D].float().view(B, T, H, D) + 3.0 * dirn
        qkv[:, H * D: 2 * H * D] = k3.reshape(B * T, H * D).to(torch.bfloat16)
        qkv[:, : H * D] = q3.reshape(B * T, H * D).to(torch.bfloat16)
    q2, k2, v2 = qkv[:, : H * D], qkv[:, H * D: 2 * H * D], qkv[:, 2 * H * D:]
    scale = D ** -0.5
    qkv_pre = qkv.clone()
    qkv_pre[:, : H * D] = (q2.float() * ops.qscale(scale)).to(torch.bfloat16)
    qp = qkv_pre[:, : H * D]
    o_qs, lse = ops.attn_fwd(qp, k2, v2, B, T, H, H, D, False, scale, want_lse=False, q_prescaled=True)
    assert lse is None
    call("slam_attn_set_fwd_qf", 60)
    try:
        o_gen, _ = ops.attn_fwd(qp, k2, v2, B, T, H, H, D, False, scale, want_lse=False, q_prescaled=True)
    finally:
        call("slam_attn_set_fwd_qf", 61)
    ref = _attn_ref(q2.float().view(B, T, H, D), k2.float().view(B, T, H, D), v2.float().view(B, T, H, D), False, None, scale)
    assert torch.isfinite(o_qs.float()).all()
    # sharply peaked rows (spikes) amplify the test's extra 2^-9 rounding of Q by the score magnitude (up to ~150 log2 units here)
    tol = 2e-2 if not grow else 6e-2
    assert_close(o_gen.view(B, T, H, D), ref, atol=tol, rtol=tol, what="attn fwd (pre-scaled Q, general softmax)")
    assert_close(o_qs.view(B, T, H, D), ref, atol=tol, rtol=tol, what="attn fwd (pre-scaled Q, accumulators from -m)")
    # same operands, same products; only where -m enters differs (before / after the fp32 accumulation)
    assert_close(o_qs.view(B, T, H, D), o_gen.view(B, T, H, D).float(), atol=1e-2, rtol=1e-2, what="accumulators from -m vs general softmax")
    # with an LSE the launch takes the general form; the LSE equals the ordinary launch's on the unscaled q up to Q's second rounding
    _, lse_pre = ops.attn_fwd(qp, k2, v2, B, T, H, H, D, False, scale, want_lse=True, q_prescaled=True)
    _, lse_ord = ops.attn_fwd(q2, k2, v2, B, T, H, H, D, False, scale, want_lse=True)
    assert_close(lse_pre[..., :T], lse_ord[..., :T], atol=(5e-2 if not grow else 0.5), rtol=1e-2, what="LSE with a pre-scaled Q")
    # and it is a deterministic function of its inputs (poisoned output buffer, second launch)
    o2 = torch.full_like(o_qs, float("nan"))
    ops.attn_fwd(qp, k2, v2, B, T, H, H, D, False, scale, want_lse=False, out=o2, q_prescaled=True)
    assert torch.equal(o2, o_qs)


@pytest.mark.parametrize("T", [100, 1500])
def test_attention_fwd_prescaled_q_tail_discards_neighbour_keys(dev, T):
    """ADVICE r5 (attention.hip QS tail): a tail tile (T % 64 != 0: Whisper's 1500 = 23 x 64 + 28) holds, past key T - 1, the first K rows
    of the NEXT batch item.  Their scores must be discarded, not merely started at -inf: with Inf / NaN in the neighbour's K rows
    (-inf + NaN = NaN) batch item 0 must come out bit-identical to the run on clean operands."""
    ops = _ops()
    B, H, D = 2, 4, 64
    qkv = rnd((B * T, 3 * H * D), dev, seed=77, std=1.0)
    scale = D ** -0.5
    qkv[:, : H * D] = (qkv[:, : H * D].float() * ops.qscale(scale)).to(torch.bfloat16)
    q2, k2, v2 = qkv[:, : H * D], qkv[:, H * D: 2 * H * D], qkv[:, 2 * H * D:]
    o_clean, _ = ops.attn_fwd(q2, k2, v2, B, T, H, H, D, False, scale, want_lse=False, q_prescaled=True)
    poisoned = qkv.clone()
    pk = poisoned[:, H * D: 2 * H * D]
    pk[T: T + 64: 2] = float("nan")          # the rows of item 1 that share item 0's last tile
    pk[T + 1: T + 64: 2] = float("inf")
    o_p, _ = ops.attn_fwd(poisoned[:, : H * D], pk, poisoned[:, 2 * H * D:], B, T, H, H, D, False, scale, want_lse=False, q_prescaled=True)
    assert torch.isfinite(o_p[:T].float()).all(), "item 0 picked up its neighbour's non-finite K rows"
    assert torch.equal(o_p[:T], o_clean[:T])


@pytest.mark.parametrize("B,T,Hq,Hkv,D,masked", [
    (2, 100, 4, 2, 64, True),
    (2, 380, 4, 1, 128, True),
    (1, 70, 2, 2, 128, False),
])
def test_attention_bwd(dev, attn_form, B, T, Hq, Hkv, D, masked):
    ops = _ops()
    qkv, q2, k2, v2, Tp, vt = _prep_attn(ops, dev, B, T, Hq, Hkv, D, seed=12, form=attn_form)
    km = None
    if masked:
        km = torch.zeros((B, Tp), dtype=torch.uint8, device=dev)
        km[:, :T] = 1
        km[0, :9] = 0
    scale = D ** -0.5
    o, lse = ops.attn_fwd(q2, k2, vt, B, T, Hq, Hkv, D, True, scale, key_mask=km)
    do = rnd((B * T, Hq * D), dev, seed=13)
    if masked:
        do.view(B, T, Hq * D)[0, :9] = 0  # pad query rows never receive gradient (labels = -100, never attended)
    dqkv = torch.zeros_like(qkv)
    dq2, dk2, dv2 = dqkv[:, : Hq * D], dqkv[:, Hq * D:(Hq + Hkv) * D], dqkv[:, (Hq + Hkv) * D:]
    ops.attn_bwd(q2, k2, v2, o, do, lse, dq2, dk2, dv2, B, T, Hq, Hkv, D, True, scale, key_mask=km)
    qf = q2.float().view(B, T, Hq, D).clone().requires_grad_(True)
    kf = k2.float().view(B, T, Hkv, D).clone().requires_grad_(True)
    vf = v2.float().view(B, T, Hkv, D).clone().requires_grad_(True)
    ref = _attn_ref(qf, kf, vf, True, km[:, :T] if masked else None, scale)
    ref.backward(do.float().view(B, T, Hq, D))
    for nme, got, r in (("dq", dq2, qf.grad), ("dk", dk2, kf.grad), ("dv", dv2, vf.grad)):
        got = got.float().reshape(r.shape)
        # cosine over the whole tensor + elementwise bound scaled to the tensor's magnitude
        cs = F.cosine_similarity(got.flatten(), r.flatten(), dim=0)
        G.floor_check(cs, 0.999, f"{nme} cosine {float(cs)}")
        assert_close(got, r, atol=3e-2 * float(r.abs().max()), rtol=3e-2, what=nme)
    # fused RoPE backward in the dQ/dK epilogues == the separate inverse-rotation pass (one rounding less)
    from slam_llm_amd.host_tables import rope_tables
    cos, sin = (t.to(dev) for t in rope_tables(T, D, 10000.0))
    fused = torch.zeros_like(qkv)
    ops.attn_bwd(q2, k2, v2, o, do, lse, fused[:, : Hq * D], fused[:, Hq * D:(Hq + Hkv) * D],
                 fused[:, (Hq + Hkv) * D:], B, T, Hq, Hkv, D, True, scale, key_mask=km, rope=(cos, sin))
    ops.head_rope_transpose(dqkv, 0, B, T, Hq, D, cos=cos, sin=sin, inverse=True, want_t=False)
    ops.head_rope_transpose(dqkv, Hq * D, B, T, Hkv, D, cos=cos, sin=sin, inverse=True, want_t=False)
    assert_close(fused, dqkv.float(), atol=2e-2 * float(dqkv.float().abs().max()), rtol=2e-2, what="fused rope grad")
    assert torch.equal(fused[:, (Hq + Hkv) * D:], dqkv[:, (Hq + Hkv) * D:])  # dV untouched by RoPE


@pytest.mark.parametrize("B,T,Hq,Hkv,D", [(2, 380, 8, 2, 128), (1, 200, 4, 1, 64), (2, 130, 4, 4, 64), (1, 97, 2, 1, 128)])
def test_attention_bwd_dq_forms_agree(dev, attn_form, B, T, Hq, Hkv, D):
    """the dQ launch forms (DMA ring, register-staged tiles with 32 / 16 queries per wave) are the same arithmetic in the
    same order: their dQ must agree to rounding noise, with left padding and RoPE"""
    ops = _ops()
    from slam_llm_amd.lib import call
    from slam_llm_amd.host_tables import rope_tables
    qkv, q2, k2, v2, Tp, vt = _prep_attn(ops, dev, B, T, Hq, Hkv, D, seed=31, form=attn_form)
    km = torch.zeros((B, Tp), dtype=torch.uint8, device=dev)
    km[:, :T] = 1
    km[0, :5] = 0
    scale = D ** -0.5
    cos, sin = (t.to(dev) for t in rope_tables(T, D, 10000.0))
    o, lse = ops.attn_fwd(q2, k2, vt, B, T, Hq, Hkv, D, True, scale, key_mask=km)
    do = rnd((B * T, Hq * D), dev, seed=32)
    do.view(B, T, Hq * D)[0, :5] = 0
    outs = {}
    try:
        for v in (0, 2, 1):
            call("slam_attn_set_bwd_variant", v)
            g = torch.zeros_like(qkv)
            ops.attn_bwd(q2, k2, v2, o, do, lse, g[:, : Hq * D], g[:, Hq * D:(Hq + Hkv) * D], g[:, (Hq + Hkv) * D:],
                         B, T, Hq, Hkv, D, True, scale, key_mask=km, rope=(cos, sin))
            outs[v] = g[:, : Hq * D].float()
    finally:
        call("slam_attn_set_bwd_variant", 0)
    ref = outs[1]
    tol = 1e-2 * float(ref.abs().max())
    for v in (0, 2):
        assert torch.isfinite(outs[v]).all()
        assert float((outs[v] - ref).abs().max()) <= tol, (v, float((outs[v] - ref).abs().max()), tol)
    assert float((outs[0] - outs[2]).abs().max()) <= 1e-3 * float(ref.abs().max())


@pytest.mark.parametrize("lens,Hq,Hkv,D", [((70, 133, 37), 4, 2, 128), ((5, 64, 1, 200, 63), 2, 2, 64), ((380, 380), 4, 1, 128)])
def test_packed_sequences_attention_equals_per_sequence(dev, attn_form, lens, Hq, Hkv, D):
    """packed ("varlen") causal attention: sequences concatenated along T with seg_lo / seg_hi == attention run on every
    sequence alone (torch fp32 reference + autograd), forward and all three gradients, incl. the fused RoPE backward
    with explicit per-token positions"""
    ops = _ops()
    from slam_llm_amd.host_tables import rope_tables
    T = sum(lens)
    qkv, q2, k2, v2, _, vt = _prep_attn(ops, dev, 1, T, Hq, Hkv, D, seed=17, form=attn_form)
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    lo = torch.tensor(np.repeat(starts, lens), dtype=torch.int32, device=dev)
    hi = torch.tensor(np.repeat(starts + np.array(lens), lens), dtype=torch.int32, device=dev)
    pos = (torch.arange(T, device=dev, dtype=torch.int32) - lo).contiguous()
    scale = D ** -0.5
    o, lse = ops.attn_fwd(q2, k2, vt, 1, T, Hq, Hkv, D, True, scale, seg=(lo, hi))
    do = rnd((T, Hq * D), dev, seed=18)
    dqkv = torch.zeros_like(qkv)
    ops.attn_bwd(q2, k2, v2, o, do, lse, dqkv[:, : Hq * D], dqkv[:, Hq * D:(Hq + Hkv) * D], dqkv[:, (Hq + Hkv) * D:],
                 1, T, Hq, Hkv, D, True, scale, seg=(lo, hi))
    ref_o = torch.empty(T, Hq, D, device=dev)
    ref_g = [torch.empty(T, Hq, D, device=dev), torch.empty(T, Hkv, D, device=dev), torch.empty(T, Hkv, D, device=dev)]
    for s0, n in zip(starts, lens):
        sl = slice(int(s0), int(s0) + n)
        qf = q2[sl].float().view(1, n, Hq, D).clone().requires_grad_(True)
        kf = k2[sl].float().view(1, n, Hkv, D).clone().requires_grad_(True)
        vf = v2[sl].float().view(1, n, Hkv, D).clone().requires_grad_(True)
        r = _attn_ref(qf, kf, vf, True, None, scale)
        r.backward(do[sl].float().view(1, n, Hq, D))
        ref_o[sl] = r[0].detach()
        for dst, src in zip(ref_g, (qf, kf, vf)):
            dst[sl] = src.grad[0]
    assert_close(o.view(T, Hq, D), ref_o, atol=2e-2, rtol=2e-2, what="packed attn fwd")
    for nme, got, r in (("dq", dqkv[:, : Hq * D], ref_g[0]), ("dk", dqkv[:, Hq * D:(Hq + Hkv) * D], ref_g[1]), ("dv", dqkv[:, (Hq + Hkv) * D:], ref_g[2])):
        got = got.float().reshape(r.shape)
        assert F.cosine_similarity(got.flatten(), r.flatten(), dim=0) > 0.999, nme
        assert_close(got, r, atol=3e-2 * float(r.abs().max()), rtol=3e-2, what="packed " + nme)
    # fused RoPE backward with per-token positions == separate inverse rotation with the same positions
    cos, sin = (t.to(dev) for t in rope_tables(max(lens), D, 10000.0))
    fused = torch.zeros_like(qkv)
    ops.attn_bwd(q2, k2, v2, o, do, lse, fused[:, : Hq * D], fused[:, Hq * D:(Hq + Hkv) * D], fused[:, (Hq + Hkv) * D:],
                 1, T, Hq, Hkv, D, True, scale, rope=(cos, sin, pos), seg=(lo, hi))
    ops.head_rope_transpose(dqkv, 0, 1, T, Hq, D, cos=cos, sin=sin, inverse=True, want_t=False, positions=pos)
    ops.head_rope_transpose(dqkv, Hq * D, 1, T, Hkv, D, cos=cos, sin=sin, inverse=True, want_t=False, positions=pos)
    assert_close(fused, dqkv.float(), atol=2e-2 * float(dqkv.float().abs().max()), rtol=2e-2, what="packed fused rope grad")


def test_cross_attention_fwd_bwd(dev, attn_form):
    """Tq != Tk (Q-Former cross-attention: 32 queries over 150 encoder frames, key padding mask), D = 64"""
    ops = _ops()
    B, Tq, Tk, H, D = 2, 32, 150, 3, 64
    q2 = rnd((B * Tq, H * D), dev, seed=21)
    kv = rnd((B * Tk, 2 * H * D), dev, seed=22)
    k2, v2 = kv[:, : H * D], kv[:, H * D:]
    vt = ops.head_rope_transpose(kv, H * D, B, Tk, H, D) if attn_form == "copies" else v2
    km = torch.zeros((B, ops.round_up(Tk, 64)), dtype=torch.uint8, device=dev)
    km[0, :Tk] = 1
    km[1, :100] = 1
    scale = D ** -0.5
    o, lse = ops.attn_fwd(q2, k2, vt, B, Tq, H, H, D, False, scale, key_mask=km, Tk=Tk)
    qf = q2.float().view(B, Tq, H, D).clone().requires_grad_(True)
    kf = k2.float().reshape(B, Tk, H, D).clone().requires_grad_(True)
    vf = v2.float().reshape(B, Tk, H, D).clone().requires_grad_(True)
    s = (qf.permute(0, 2, 1, 3) @ kf.permute(0, 2, 3, 1)) * scale
    s = s.masked_fill(~km[:, None, None, :Tk].bool(), float("-inf"))
    ref = (torch.softmax(s, -1) @ vf.permute(0, 2, 1, 3)).permute(0, 2, 1, 3)
    assert_close(o.view(B, Tq, H, D), ref.detach(), atol=2e-2, rtol=2e-2, what="cross attn fwd")
    do = rnd((B * Tq, H * D), dev, seed=23)
    dq2 = torch.zeros_like(q2)
    dkv = torch.zeros_like(kv)
    ops.attn_bwd(q2, k2, v2, o, do, lse, dq2, dkv[:, : H * D], dkv[:, H * D:], B, Tq, H, H, D, False, scale,
                 key_mask=km, Tk=Tk)
    ref.backward(do.float().view(B, Tq, H, D))
    for nme, got, r in (("dq", dq2, qf.grad), ("dk", dkv[:, : H * D], kf.grad), ("dv", dkv[:, H * D:], vf.grad)):
        got = got.float().reshape(r.shape)
        cs = F.cosine_similarity(got.flatten(), r.flatten(), dim=0)
        G.floor_check(cs, 0.9995, f"{nme} cosine {float(cs)}")
        assert_close(got, r, atol=3e-2 * float(r.abs().max()), rtol=3e-2, what=nme)


@pytest.mark.parametrize("lens,H,D", [((70, 133, 37), 4, 64), ((5, 64, 1, 200, 63), 2, 64), ((1500, 431, 1500), 2, 64), ((130, 70), 2, 128)])
def test_packed_bidirectional_attention_equals_per_clip(dev, attn_form, lens, H, D):
    """the ragged encoder's attention: clips concatenated along T (B = 1), query q sees keys seg_lo[q] <= k < seg_hi[q] and
    nothing else == bidirectional attention run on every clip alone (fp32 torch reference).  Covers clips shorter than one
    64-key tile, boundaries in the middle of tiles and of 16-row fragments, and full 1500-frame clips."""
    ops = _ops()
    T = sum(lens)
    qkv, q2, k2, v2, _, vt = _prep_attn(ops, dev, 1, T, H, H, D, seed=31, form=attn_form)
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    lo = torch.tensor(np.repeat(starts, lens), dtype=torch.int32, device=dev)
    hi = torch.tensor(np.repeat(starts + np.array(lens), lens), dtype=torch.int32, device=dev)
    scale = D ** -0.5
    o, _ = ops.attn_fwd(q2, k2, vt, 1, T, H, H, D, False, scale, want_lse=False, seg=(lo, hi))
    o = o.float().view(T, H, D)
    for s0, n in zip(starts, lens):
        qf = q2[s0:s0 + n].float().view(1, n, H, D)
        kf = k2[s0:s0 + n].float().view(1, n, H, D)
        vf = v2[s0:s0 + n].float().view(1, n, H, D)
        ref = _attn_ref(qf, kf, vf, False, None, scale)[0]
        assert_close(o[s0:s0 + n], ref, atol=2e-2, rtol=2e-2, what=f"clip at {s0} len {n}")


def test_gather_rows_pack_unpack_and_windows(dev):
    """slam_gather_rows_bf16: bit-exact row gather (pack), inverse index with -1 -> zero rows (un-pack), and windows wider than
    the source row (the projector's k-frame stack over a packed encoder output)"""
    ops = _ops()
    src = rnd((301, 136), dev, seed=3)[:, :128]                 # strided source view
    idx = torch.tensor([5, 0, 300, 17, 17, 299], dtype=torch.int32, device=dev)
    assert torch.equal(ops.gather_rows(src, idx), src[idx.long()])
    inv = torch.tensor([2, -1, 0, -1, 1], dtype=torch.int32, device=dev)
    got = ops.gather_rows(src, inv)
    assert torch.equal(got[[0, 2, 4]], src[[2, 0, 1]]) and float(got[[1, 3]].abs().max()) == 0.0
    enc = rnd((97, 64), dev, seed=4)                             # contiguous: windows of k = 5 rows
    win = torch.tensor([0, 5, 10, 23, 28, 92], dtype=torch.int32, device=dev)
    got = ops.gather_rows(enc, win, width=5 * 64)
    for r, w0 in enumerate(win.tolist()):
        assert torch.equal(got[r], enc[w0:w0 + 5].reshape(-1))


@pytest.mark.parametrize("stride,C,dt", [(1, 80, torch.float32), (2, 128, torch.bfloat16)])
def test_conv_im2col_respects_per_clip_lengths(dev, stride, C, dt):
    """n_valid: frames at or past a clip's own length read as zero -- equal to running the im2col on a copy whose tail was
    zeroed, whatever garbage (here: NaN) sits in the pad rows"""
    ops = _ops()
    B, T = 3, 50
    x = rnd((B, T, C), dev, seed=6, dtype=dt)
    nv = torch.tensor([50, 13, 36], dtype=torch.int32, device=dev)
    clean = x.clone()
    dirty = x.clone()
    for b, n in enumerate(nv.tolist()):
        clean[b, n:] = 0
        dirty[b, n:] = float("nan")
    Kp = ops.round_up(3 * C, 64)
    assert torch.equal(ops.conv1d_k3_im2col(dirty, stride, Kp, n_valid=nv), ops.conv1d_k3_im2col(clean, stride, Kp))


# ----------------------------------------------------------------------------------------- mlp / conv
def test_swiglu(dev):
    ops = _ops()
    M, Fd = 123, 256
    gu = rnd((M, 2 * Fd), dev, seed=1, std=2.0)
    dh = rnd((M, Fd), dev, seed=2)
    h = ops.swiglu_fwd(gu)
    g = gu[:, :Fd].float().requires_grad_(True)
    u = gu[:, Fd:].float().requires_grad_(True)
    ref = F.silu(g) * u
    assert_close(h, ref.detach(), atol=1e-2, rtol=1e-2, what="swiglu fwd")
    ref.backward(dh.float())
    dgu = ops.swiglu_bwd(gu, dh)
    assert_close(dgu[:, :Fd], g.grad, atol=2e-2, rtol=1.5e-2, what="swiglu dgate")
    assert_close(dgu[:, Fd:], u.grad, atol=2e-2, rtol=1.5e-2, what="swiglu dup")


@pytest.mark.parametrize("stride,C,dt", [(1, 80, torch.float32), (2, 64, torch.bfloat16)])
def test_conv1d_as_gemm(dev, stride, C, dt):
    ops = _ops()
    B, Tin, Cout = 2, 101, 128
    x = rnd((B, Tin, C), dev, seed=1, dtype=dt)
    w = rnd((Cout, C, 3), dev, seed=2, std=0.1, dtype=torch.float32)
    bias = rnd((Cout,), dev, seed=3, dtype=torch.float32)
    Kp = ops.round_up(3 * C, 64)
    cols = ops.conv1d_k3_im2col(x, stride, Kp)
    w2 = torch.zeros((Cout, Kp), dtype=torch.bfloat16, device=dev)
    w2[:, : 3 * C] = w.permute(0, 2, 1).reshape(Cout, 3 * C).to(torch.bfloat16)
    y = ops.gemm_nt(cols, w2, bias=bias, act=ops.ACT_GELU)
    ref = F.gelu(F.conv1d(x.float().permute(0, 2, 1), w.to(torch.bfloat16).float(), bias, stride=stride, padding=1))
    ref = ref.permute(0, 2, 1).reshape(-1, Cout)
    assert y.shape == ref.shape
    assert_close(y, ref, atol=2e-2, rtol=1e-2, what="conv")


# ----------------------------------------------------------------------------------------- embed / loss / optim
def test_embed_splice(dev):
    ops = _ops()
    B, T, Ta, d, V = 3, 40, 12, 128, 200
    E = rnd((V, d), dev, seed=1)
    enc = rnd((B, Ta, d), dev, seed=2)
    ids = torch.randint(1, V, (B, T), dtype=torch.int64)
    mask = torch.zeros((B, T), dtype=torch.bool)
    mask[0, 5:17] = True          # exactly Ta
    mask[1, 0:14] = True          # longer than Ta -> clamp, slots 12,13 get zeros
    mask[2, 20:25] = True         # shorter
    ids[mask] = -1
    ids_d = ids.clone().to(dev)
    out, spans = ops.embed_splice_fwd(ids_d, mask.to(torch.uint8).to(dev), E, enc)
    # reference (slam_model.py:370-392)
    from oracle.slam_oracle import embed_splice
    ref = embed_splice(E.float().cpu(), ids.clone(), mask, enc.float().cpu())
    assert torch.equal(out.view(B, T, d).float().cpu(), ref)
    assert (ids_d >= 0).all()
    dx = rnd((B * T, d), dev, seed=3)
    de = ops.embed_splice_bwd(spans, dx, B, T, Ta, d).view(B, Ta, d).cpu()
    dxc = dx.view(B, T, d).cpu()
    assert torch.equal(de[0], dxc[0, 5:17])
    assert torch.equal(de[1], dxc[1, 0:12])
    assert torch.equal(de[2, :5], dxc[2, 20:25]) and torch.all(de[2, 5:] == 0)


@pytest.mark.parametrize("V", [512, 32000])
def test_cross_entropy(dev, V):
    ops = _ops()
    B, T = 3, 17
    logits = rnd((B * T, V), dev, seed=1, std=2.0)
    labels = torch.randint(0, V, (B, T), dtype=torch.int64)
    labels[:, :6] = -100
    labels[1, 12:] = -100
    labels_d = labels.to(dev)
    tgt, nv = ops.ce_targets(labels_d)
    row_loss = torch.empty(B * T, dtype=torch.float32, device=dev)
    row_ok = torch.empty(B * T, dtype=torch.int32, device=dev)
    lg = logits.clone()
    ops.ce_fwd_bwd(lg, tgt, nv, row_loss, row_ok, write_grad=True)
    out = ops.ce_finalize(row_loss, row_ok, nv).cpu()
    lf = logits.float().cpu().view(B, T, V).requires_grad_(True)
    sl = F.pad(labels, (0, 1), value=-100)[:, 1:]
    ref = F.cross_entropy(lf.view(-1, V), sl.reshape(-1), ignore_index=-100)
    ref.backward()
    assert int(nv.item()) == int((sl != -100).sum())
    assert abs(float(out[0]) - float(ref)) < 1e-4 * max(1.0, abs(float(ref)))
    preds = lf.detach().argmax(-1)
    m = sl != -100
    acc = (preds[m] == sl[m]).float().mean()
    assert abs(float(out[1]) - float(acc)) < 1e-6
    assert_close(lg.view(B, T, V), lf.grad, atol=2e-5, rtol=1.5e-2, what="dlogits")


def test_adamw_matches_torch(dev):
    ops = _ops()
    n = 10007
    p0 = torch.randn(n)
    p = p0.clone().to(dev)
    m = torch.zeros(n, device=dev)
    v = torch.zeros(n, device=dev)
    pb = torch.empty(n, dtype=torch.bfloat16, device=dev)
    tp = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([tp], lr=1e-2, weight_decay=0.01)
    for step in range(1, 4):
        g = torch.randn(n, generator=torch.Generator().manual_seed(step))
        tp.grad = g.clone()
        opt.step()
        ops.adamw_step(p, g.to(dev), m, v, pb, 1e-2, 0.9, 0.999, 1e-8, 0.01, step)
    assert_close(p, tp.detach(), atol=1e-6, rtol=1e-5, what="adamw params")
    assert torch.equal(pb.cpu(), p.cpu().to(torch.bfloat16))


# ----------------------------------------------------------------------------------------- log-mel
@pytest.mark.parametrize("n_mels", [80, 128])
def test_logmel_vs_golden_and_oracle(dev, n_mels):
    ops = _ops()
    from oracle import slam_oracle as O
    fx = G.load("logmel")
    audio = torch.from_numpy(fx["audio"]).to(dev)  # 3.7 s clips, padded to 30 s inside the kernel
    mel = ops.logmel(audio, n_mels).cpu()  # [2, 3000, n_mels]
    idx = fx[f"mel{n_mels}_frames"]
    gold = torch.from_numpy(fx[f"mel{n_mels}_values"]).permute(0, 2, 1)  # [2, nidx, n_mels]
    # SURVEY 8c tolerance: mel fp32 abs <= 1e-4
    assert_close(mel[:, idx, :], gold, atol=1e-4, rtol=0, what="logmel vs reference fixture")
    full = torch.stack([O.log_mel_spectrogram(O.pad_or_trim(a), n_mels) for a in audio.cpu()]).permute(0, 2, 1)
    assert_close(mel, full, atol=1e-4, rtol=0, what="logmel vs oracle")


@pytest.mark.parametrize("n_mels,secs", [(20, 1.0), (100, 2.3), (256, 1.5), (128, 30.0)])
def test_logmel_other_filter_counts_and_lengths(dev, n_mels, secs):
    """the round-4 kernel's thread -> (filter, frames) mapping at filter counts that do not divide its 512 threads (100), need two
    8-frame passes per thread (256) or leave most of a frame group idle (20), on clips whose frame count is not a multiple of the
    32-frame block, and one full 30 s clip (94 blocks walked by one workgroup chain); against the oracle, same 1e-4 tolerance"""
    ops = _ops()
    from oracle import slam_oracle as O
    n = int(secs * 16000) // 160 * 160
    g = torch.Generator().manual_seed(n_mels)
    audio = (torch.randn(3, n, generator=g) * 0.1).clamp(-1, 1)
    audio[1, n // 2:] = 0.0                      # a silent tail: the clamp at 1e-10 and the max - 8 floor are live
    mel = ops.logmel(audio.to(dev), n_mels, n_samples=n).cpu()
    ref = torch.stack([O.log_mel_spectrogram(a, n_mels) for a in audio]).permute(0, 2, 1)
    assert mel.shape == ref.shape == (3, n // 160, n_mels)
    assert_close(mel, ref, atol=1e-4, rtol=0, what=f"logmel n_mels={n_mels}")


def test_logmel_per_clip_ragged(dev):
    """pad_or_trim off: each clip's own STFT/floor, mel-space zero padding to the batch maximum (speech_dataset_large.py)"""
    ops = _ops()
    from oracle import slam_oracle as O
    lens = [16000 * 2 + 37, 16000 * 3, 9000]
    g = torch.Generator().manual_seed(3)
    clips = [(torch.randn(n, generator=g) * 0.1).clamp(-1, 1) for n in lens]
    nmax = max(lens)
    audio = torch.stack([torch.nn.functional.pad(c, (0, nmax - len(c))) for c in clips]).to(dev)
    nv = torch.tensor(lens, dtype=torch.int32, device=dev)
    mel = ops.logmel(audio, 80, n_samples=ops.round_up(nmax, 160), n_valid=nv, per_clip=True).cpu()
    for i, c in enumerate(clips):
        ref = O.log_mel_spectrogram(c, 80).permute(1, 0)  # [frames_i, 80] over the clip's own length
        fi = ref.shape[0]
        assert fi == lens[i] // 160
        assert_close(mel[i, :fi], ref, atol=1e-4, rtol=0, what=f"ragged clip {i}")
        assert torch.all(mel[i, fi:] == 0)


def test_dropout_mask_properties(dev):
    ops = _ops()
    M, N, p = 500, 256, 0.25
    x = rnd((M, N), dev, seed=1)
    y1 = ops.dropout(x, p, seed=123, offset=1 << 40)
    y2 = ops.dropout(x, p, seed=123, offset=1 << 40)
    y3 = ops.dropout(x, p, seed=123, offset=2 << 40)
    assert torch.equal(y1, y2) and not torch.equal(y1, y3)          # pure function of (seed, offset, index)
    ones = torch.ones((M, N), dtype=torch.bfloat16, device=dev)
    mask = ops.dropout(ones, p, seed=123, offset=1 << 40).float()
    # the device mask IS the host restatement of the hash (tests/golden_util.mix64), bit for bit, also for a seed with high bits set and an
    # offset whose group index crosses 2^32
    for seed_, off_ in ((123, 1 << 40), (2 ** 63 + 99, (7 << 40) + 8), (0, (1 << 34) - 64)):
        dm = ops.dropout(ones, p, seed=seed_, offset=off_).float().cpu().ne(0)
        idx = np.uint64(off_) + np.arange(M * N, dtype=np.uint64)
        z = G.mix64(seed_, idx >> np.uint64(2))
        bits = (z >> (np.uint64(16) * (idx & np.uint64(3)))) & np.uint64(0xFFFF)
        host = torch.from_numpy((bits >= np.uint64(min(65535, int(p * 65536.0 + 0.5)))).reshape(M, N))
        assert torch.equal(dm, host), (seed_, off_)
    keep = (mask > 0).float().mean().item()
    assert abs(keep - (1 - p)) < 0.01, keep
    assert torch.allclose(mask[mask > 0], torch.tensor(1 / (1 - p)), rtol=1e-2)
    assert_close(y1, x.float() * mask, atol=1e-2, rtol=1e-2, what="dropout scaling")
    acc = x.clone()
    ops.dropout(x, p, seed=123, offset=1 << 40, out=acc, accumulate=True)
    assert_close(acc, x.float() + x.float() * mask, atol=2e-2, rtol=1e-2, what="dropout accumulate")
    assert torch.equal(ops.dropout(x, 0.0, seed=5, offset=0), x)


@pytest.mark.parametrize("M,K,R,p", [(200, 128, 16, 0.3), (1000, 4096, 32, 0.05), (77, 256, 24, 0.0), (33, 64, 64, 0.5)])
def test_lora_first_hop_and_gram_recompute_the_dropout_mask(dev, M, K, R, p):
    """u = dropout(x) A^T and dA = du^T dropout(x) with the mask recomputed in registers == the same products on the
    materialised ops.dropout(x) (same seed / offset / element index)"""
    ops = _ops()
    x, A = rnd((M, K), dev, seed=31), rnd((R, K), dev, seed=32, std=K ** -0.5)
    drop = (p, 987654321, 3 << 40) if p > 0 else None
    xd = ops.dropout(x, *drop) if drop else x
    u = torch.empty((M, R + 8), dtype=torch.bfloat16, device=dev)
    ops.lora_a_fwd(x, A, u[:, :R], drop)
    assert_close(u[:, :R], xd.float() @ A.float().T, atol=2e-2, rtol=2e-2, what="lora first hop")
    # pad_to (round 5): the padding columns of the K-extension behind the results are zeroed by the same launch -- and nothing else moves
    Rp = ops.round_up(R, 64)
    up = torch.full((M, Rp + 8), float("nan"), dtype=torch.bfloat16, device=dev)
    ops.lora_a_fwd(x, A, up[:, :R], drop, pad_to=Rp)
    assert torch.equal(up[:, :R], u[:, :R])
    assert torch.all(up[:, R:Rp] == 0) and torch.isnan(up[:, Rp:].float()).all()
    if R in (8, 16, 32, 64):
        du = rnd((M, R), dev, seed=33)
        g1 = torch.zeros((R, K), dtype=torch.float32, device=dev)
        ops.skinny_gram(du, x, g1, K, 1, drop=drop)
        ref = du.float().T @ xd.float()
        assert_close(g1, ref, atol=2e-2 * float(ref.abs().max()), rtol=2e-2, what="dA with recomputed mask")


@pytest.mark.parametrize("M,K,R,sr", [(700, 512, 64, 32), (11780, 4096, 64, 32), (100, 256, 64, 8), (380, 2048, 64, 16), (130, 264, 64, 16), (3000, 1096, 64, 64)])
def test_lora_hop_dropout_equals_the_two_launch_form(dev, M, K, R, sr):
    """slam_lora_hop_dropout (second hop of the LoRA backward + recomputed dropout mask + accumulate, one pass over dx) is BIT-identical to
    the product into a scratch buffer followed by slam_dropout_bf16(accumulate) -- same MFMA order, same two bf16 roundings, same mask"""
    ops = _ops()
    du = torch.zeros((M, R), dtype=torch.bfloat16, device=dev)
    du[:, :sr] = rnd((M, sr), dev, seed=101)
    a_t = torch.zeros((K, R), dtype=torch.bfloat16, device=dev)
    a_t[:, :sr] = rnd((K, sr), dev, seed=102, std=0.05)
    dx0 = rnd((M, K), dev, seed=103)
    drop = (0.05, 1234567, 5 << 40)
    ref = dx0.clone()
    ops.dropout(ops.gemm_nt(du, a_t), *drop, out=ref, accumulate=True)
    got = dx0.clone()
    ops.lora_hop_dropout(du, a_t, got, drop)
    assert torch.equal(got, ref)
    assert not torch.equal(got, dx0)
    # a column view with a larger leading dimension (how FusedLinear.backward calls it: du and dx are column blocks of dx_ext)
    ext = torch.zeros((M, K + R), dtype=torch.bfloat16, device=dev)
    ext[:, :K], ext[:, K:] = dx0, du
    ops.lora_hop_dropout(ext[:, K:], a_t, ext[:, :K], drop)
    assert torch.equal(ext[:, :K], ref)


def test_lora_fused_linear_with_dropout_matches_autograd(dev):
    """peft semantics y = x W^T + s * (dropout(x) A^T) B^T, forward and every gradient, with the kernel's own mask"""
    ops = _ops()
    from slam_llm_amd.model import FusedLinear, TrainableStore
    M, K, r, alpha, p = 200, 128, 8, 32.0, 0.3
    store = TrainableStore(dev)
    fl = FusedLinear(K, [("q_proj", 128), ("k_proj", 64), ("v_proj", 64)], dev)
    for n, rows in (("q_proj", 128), ("v_proj", 64)):
        store.reserve(f"{n}.A", (r, K))
    for n, rows in (("q_proj", 128), ("v_proj", 64)):
        store.reserve(f"{n}.B", (rows, r))
        fl.add_lora(n, r, alpha, f"{n}.A", f"{n}.B")
    store.allocate()
    g = torch.Generator().manual_seed(0)
    Wb = {n: (torch.randn(rows, K, generator=g) * K ** -0.5) for n, rows in fl.parts}
    for n, rows in fl.parts:
        fl.set_base(n, Wb[n].to(dev).to(torch.bfloat16))
    fl.finalize()
    with torch.no_grad():
        for n, prm in store.params.items():
            prm.copy_((torch.randn(prm.shape, generator=g) * 0.1).to(dev))
    store.refresh_bf16()
    fl.refresh(store)
    x_ext = fl.new_input(M)
    x = (torch.randn(M, K, generator=g)).to(torch.bfloat16)
    x_ext[:, :K] = x.to(dev)
    drop = (p, 77, 5 << 40)
    y = fl.forward(x_ext, store, drop=drop)
    mask = ops.dropout(torch.ones((M, K), dtype=torch.bfloat16, device=dev), *drop).float().cpu()
    # torch reference (bf16-rounded operands, fp32 math)
    xf = x.float().requires_grad_(True)
    Wcat = torch.cat([Wb[n].to(torch.bfloat16).float() for n, _ in fl.parts], 0)
    P = {n: store.params[n].detach().cpu().to(torch.bfloat16).float().requires_grad_(True) for n in store.params}
    xd = xf * mask
    yq = xf @ Wcat[:128].T + (alpha / r) * (xd @ P["q_proj.A"].T) @ P["q_proj.B"].T
    yk = xf @ Wcat[128:192].T
    yv = xf @ Wcat[192:].T + (alpha / r) * (xd @ P["v_proj.A"].T) @ P["v_proj.B"].T
    ref = torch.cat([yq, yk, yv], 1)
    assert_close(y, ref.detach(), atol=3e-2, rtol=2e-2, what="lora+dropout fwd")
    dy = rnd((M, 256), dev, seed=9)
    ref.backward(dy.float().cpu())
    dx_ext = fl.backward(dy, x_ext, store, accumulate=False, drop=drop)
    assert F.cosine_similarity(dx_ext[:, :K].float().cpu().flatten(), xf.grad.flatten(), dim=0) > 0.9995
    assert_close(dx_ext[:, :K], xf.grad, atol=3e-2 * float(xf.grad.abs().max()), rtol=3e-2, what="dx")
    for n in store.params:
        gr = store.grad_view(n).float().cpu()
        cs = F.cosine_similarity(gr.flatten(), P[n].grad.flatten(), dim=0)
        G.floor_check(cs, 0.999, f"{n}: cosine {float(cs)}")


# ----------------------------------------------------------------------------------------- decode kernels
@pytest.mark.parametrize("M", [1, 5, 16, 17, 40, 64])
@pytest.mark.parametrize("N,K,splits", [(96, 64, 0), (4128, 4160, 0), (1000, 512, 1), (256, 1024, 3), (50176, 256, 0)])
def test_gemm_skinny_matches_fp32_matmul(dev, M, N, K, splits):
    ops = _ops()
    a, b = rnd((M, K), dev, seed=3), rnd((N, K), dev, seed=4)
    res = rnd((M, N), dev, seed=5)
    ref = a.float() @ b.float().T
    ops.SKINNY_SPLITS = splits
    try:
        c32 = ops.gemm_nt(a, b, out_dtype=torch.float32)
        cbf = ops.gemm_nt(a, b, residual=res)
        again = ops.gemm_nt(a, b, out_dtype=torch.float32)
    finally:
        ops.SKINNY_SPLITS = 0
    # fp32 products and accumulation: only the summation order differs from torch's
    assert_close(c32, ref, atol=2e-4 * math.sqrt(K), rtol=1e-5, what=f"skinny f32 {M}x{N}x{K}")
    assert_close(cbf, ref + res.float(), atol=2e-2 * math.sqrt(K / 64), rtol=1e-2, what=f"skinny bf16+res {M}x{N}x{K}")
    assert torch.equal(c32, again), "split-K reduction must be bit-reproducible"


def test_gemm_skinny_asymmetric_identity(dev):
    """x = rows of I with an asymmetric W catches row/col or k-permutation mistakes in the MFMA mapping."""
    ops = _ops()
    K = 256
    a = torch.eye(K, dtype=torch.bfloat16, device=dev)[200:248]
    b = (torch.arange(200 * K, dtype=torch.float32).reshape(200, K) % 251 - 125).to(torch.bfloat16).to(dev)
    c = ops.gemm_nt(a, b, out_dtype=torch.float32)
    assert torch.equal(c.cpu(), b.float().T[200:248].cpu())
    c16 = ops.gemm_nt(a[:13], b, out_dtype=torch.float32)   # M <= 16 kernel (in-workgroup split-K)
    assert torch.equal(c16.cpu(), b.float().T[200:213].cpu())


def test_gemm_skinny_stacked_weights_and_swiglu(dev):
    """two-block weight rows (LoRA A stacked under W) and the fused SwiGLU epilogue, both kernels (M <= 16 / <= 64)"""
    ops = _ops()
    K, N1, N2, Fd = 512, 200, 24, 1088
    w, w2, wg = rnd((N1, K + 64), dev, seed=21), rnd((N2, K), dev, seed=22), rnd((2 * Fd, K), dev, seed=23, std=0.05)
    for M in (3, 16, 40):
        x = rnd((M, K), dev, seed=24)
        out = torch.empty((M, N1 + N2), dtype=torch.bfloat16, device=dev)
        ops.gemm_skinny(x, w[:, :K], out, b2=w2)
        ref = torch.cat([x.float() @ w[:, :K].float().T, x.float() @ w2.float().T], dim=1)
        assert_close(out, ref, atol=2e-2 * math.sqrt(K / 64), rtol=1e-2, what=f"stacked rows M={M}")
        for splits in (0, 2):
            ops.SKINNY_SPLITS = splits
            try:
                hh = torch.empty((M, Fd), dtype=torch.bfloat16, device=dev)
                ops.gemm_skinny(x, wg, hh, swiglu=True)
            finally:
                ops.SKINNY_SPLITS = 0
            gu = (x.float() @ wg.float().T).to(torch.bfloat16)
            assert torch.equal(hh, ops.swiglu_fwd(gu)) or (hh.float() - ops.swiglu_fwd(gu).float()).abs().max() < 2e-2, \
                f"fused swiglu M={M} splits={splits}"


@pytest.mark.parametrize("D,Hq,Hkv,lora", [(128, 8, 2, 16), (64, 4, 4, 0), (64, 8, 1, 12), (128, 4, 2, 0)])
def test_decode_attention_fused(dev, D, Hq, Hkv, lora):
    """one decode step (LoRA delta + RoPE + KV append + attention in one launch) against a torch fp32 reference that
    materialises each hypothesis' history from the ancestor table; prompt KV shared by the beams of an item, ragged
    left padding."""
    ops = _ops()
    from slam_llm_amd.host_tables import rope_tables
    B, beams, T, G, n = 2, 3, 37, 9, 5   # n generated tokens already cached
    R, HD, NQ = B * beams, Hkv * D, (Hq + 2 * Hkv) * D
    g = torch.Generator().manual_seed(11)
    ld = NQ + (lora + 7) // 8 * 8
    qkv = rnd((R, ld), dev, seed=6)
    lora_b = rnd((NQ, 16), dev, seed=12, std=0.3) if lora else None
    Kp, Vp = rnd((B, T, HD), dev, seed=7), rnd((B, T, HD), dev, seed=8)
    Kg, Vg = rnd((R, G, HD), dev, seed=9), rnd((R, G, HD), dev, seed=10)
    start = torch.tensor([0, 6], dtype=torch.int32, device=dev)
    anc = torch.stack([torch.randint(0, beams, (G,), generator=g) + (r // beams) * beams for r in range(R)]).to(torch.int32).to(dev)
    positions = torch.tensor([T - int(start[r // beams]) + n for r in range(R)], dtype=torch.int32, device=dev)
    cos, sin = rope_tables(T + G, D, 10000.0)
    cos, sin = cos.to(dev), sin.to(dev)
    Kg0, Vg0 = Kg.clone(), Vg.clone()
    out = torch.empty((R, Hq * D), dtype=torch.bfloat16, device=dev)
    ops.attn_decode(qkv, lora_b, lora, cos, sin, positions, Kp, Vp, start, Kg, Vg, anc, None, n, beams, Hq, Hkv, D,
                    D ** -0.5, out)

    bf = lambda t: t.to(torch.bfloat16).float()  # noqa: E731
    y = qkv[:, :NQ].float()
    if lora:
        y = bf(y + bf(qkv[:, NQ:NQ + lora].float() @ lora_b[:, :lora].float().T))

    def rope(x, pos):   # x [R, H, D] fp32, HF rotate_half convention; rounded to bf16 like the kernel's stores
        c = torch.cat([cos[pos.long()], cos[pos.long()]], -1)[:, None, :].float()
        s = torch.cat([sin[pos.long()], sin[pos.long()]], -1)[:, None, :].float()
        rot = torch.cat([-x[..., D // 2:], x[..., : D // 2]], -1)
        return bf(x * c + rot * s)

    qf = rope(y[:, : Hq * D].view(R, Hq, D), positions)
    kf = rope(y[:, Hq * D:(Hq + Hkv) * D].view(R, Hkv, D), positions)
    vf = y[:, (Hq + Hkv) * D:].view(R, Hkv, D)
    assert_close(Kg[:, n].view(R, Hkv, D), kf, atol=1e-6, rtol=2 ** -7, what="K append (<= 1 bf16 ulp: fma order)")
    assert_close(Vg[:, n].view(R, Hkv, D), vf, atol=1e-6, rtol=2 ** -7, what="V append")
    keep = torch.ones(G, dtype=torch.bool)
    keep[n] = False
    assert torch.equal(Kg[:, keep], Kg0[:, keep]) and torch.equal(Vg[:, keep], Vg0[:, keep]), "other slots untouched"
    assert torch.equal(anc[:, n].cpu(), torch.arange(R, dtype=torch.int32))
    ref = torch.empty(R, Hq, D)
    rep = Hq // Hkv
    for r in range(R):
        it, s0 = r // beams, int(start[r // beams])
        rows = anc[r, : n + 1].long()
        Kh = torch.cat([Kp[it, s0:].float(), Kg[rows, torch.arange(n + 1, device=dev)].float()]).view(-1, Hkv, D).cpu()
        Vh = torch.cat([Vp[it, s0:].float(), Vg[rows, torch.arange(n + 1, device=dev)].float()]).view(-1, Hkv, D).cpu()
        for hq in range(Hq):
            sc = (Kh[:, hq // rep] @ qf[r, hq].cpu()) * D ** -0.5
            ref[r, hq] = torch.softmax(sc, 0) @ Vh[:, hq // rep]
    assert_close(out.view(R, Hq, D), ref, atol=2e-2, rtol=1e-2, what="decode attention")


@pytest.mark.parametrize("name", ["fp32_params", "bf16_params", "bf16_params_kahan"])
def test_anyprecision_adamw_matches_reference_class_fixture(dev, name):
    """fused AnyPrecisionAdamW kernel, 6 steps, against
      (a) the oracle restatement executed with torch ops ON THE DEVICE -- the platform the reference's optimizer runs on; the
          restatement itself is pinned bit for bit to the reference's own class by tests/test_oracle_golden.py -- expected equal up
          to one bf16 ulp on a handful of elements (fma contraction of a + alpha*b);
      (b) for fp32 parameters also tests/golden/anyprecision.npz (written by the reference class on the CPU).  The bf16-parameter
          fixtures are NOT comparable on a GPU: torch's CPU kernel rounds `alpha` of `add_(bf16, alpha=)` to bf16 first (0.1 ->
          0.10009765625: 20 % of the momenta move by one ulp), the device kernel keeps it in fp32 -- as does this kernel."""
    ops = _ops()
    from oracle import slam_oracle as O
    from tests.test_oracle_golden import ANYPRECISION_CASES
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "anyprecision.npz"))
    pdt, kahan, wd = ANYPRECISION_CASES[name]
    pbf = pdt == torch.bfloat16
    p = torch.from_numpy(fx["p0"]).to(dev)
    if pbf:
        p = p.to(torch.bfloat16).float()
    p_ref = p.to(pdt).clone()
    st_ref = {}
    n = p.numel()
    m = torch.zeros(n, dtype=torch.bfloat16, device=dev)
    v = torch.zeros(n, dtype=torch.bfloat16, device=dev)
    c = torch.zeros(n, dtype=torch.bfloat16, device=dev) if kahan else None
    pb = torch.empty(n, dtype=torch.bfloat16, device=dev)

    def ulps(got, want):      # distance in bf16 ulps
        a = got.float().cpu().view(torch.int32) >> 16
        b = want.float().cpu().view(torch.int32) >> 16
        return (a - b).abs()
    for s in range(6):
        g = torch.from_numpy(fx[f"grad.{s}"]).to(dev)
        ops.adamw_anyprecision_step(p, g, m, v, c, pb, float(fx["lr"]), 0.9, 0.999, 1e-8, wd, s + 1, params_are_bf16=pbf)
        O.anyprecision_adamw_step(p_ref, g.to(pdt), st_ref, float(fx["lr"]), weight_decay=wd, use_kahan_summation=kahan)
        pairs = [("m", m, st_ref["exp_avg"]), ("v", v, st_ref["exp_avg_sq"])] + ([("c", c, st_ref["compensation"])] if kahan else [])
        for nme, t, want in pairs:
            d = ulps(t, want)
            assert float((d > 0).float().mean()) < 0.02, (name, nme, s, float((d > 0).float().mean()))
            if nme == "c":   # the compensation is the rounding residual of p: a 1-ulp difference upstream moves it by up to one ulp OF p
                assert bool(((t.float() - want.float()).abs() <= 2.0 ** -7 * p_ref.float().abs() + 1e-12).all()), (name, s)
            else:
                assert int(d.max()) <= 1, (name, nme, s, int(d.max()))
        err = (p - p_ref.float()).abs().cpu()
        tol = (2.0 ** -7 if pbf else 1e-5) * p_ref.float().abs().clamp(min=1e-3).cpu()
        assert float((err > tol).float().mean()) < 0.02, (name, s, float(err.max()))
        assert torch.equal(pb.float().cpu(), p.cpu().to(torch.bfloat16).float())
        if not pbf:
            want = torch.from_numpy(fx[f"{name}.p.{s}"])
            assert float(((p.cpu() - want).abs() > 1e-5 * want.abs().clamp(min=1e-3)).float().mean()) < 0.02
            assert float((ulps(m, torch.from_numpy(fx[f"{name}.m.{s}"])) > 0).float().mean()) < 0.02


@pytest.mark.parametrize("stride,T", [(1, 37), (2, 37), (2, 40)])
def test_conv_col2im_is_the_adjoint_of_im2col(dev, stride, T):
    """slam_conv1d_k3_col2im vs autograd through an unfold-style im2col (fp32): <im2col(x), c> == <x, col2im(c)>"""
    from slam_llm_amd import ops
    B, C = 3, 64
    g = torch.Generator().manual_seed(5)
    To = (T + 2 - 3) // stride + 1
    dcols = torch.randn(B * To, 3 * C, generator=g).to(torch.bfloat16)
    x = torch.zeros(B, T, C, requires_grad=True)
    xp = torch.nn.functional.pad(x, (0, 0, 1, 1))
    cols = torch.cat([xp[:, j: j + stride * (To - 1) + 1: stride] for j in range(3)], dim=-1).reshape(B * To, 3 * C)
    (cols * dcols.float()).sum().backward()
    got = ops.conv1d_k3_col2im(dcols.to(dev), B, T, C, stride).float().cpu()
    assert torch.allclose(got, x.grad, atol=2e-2, rtol=1e-2), (got - x.grad).abs().max()
    # and im2col itself agrees with the same unfold
    xr = torch.randn(B, T, C, generator=g).to(torch.bfloat16)
    xpr = torch.nn.functional.pad(xr.float(), (0, 0, 1, 1))
    ref = torch.cat([xpr[:, j: j + stride * (To - 1) + 1: stride] for j in range(3)], dim=-1).reshape(B * To, 3 * C)
    assert torch.equal(ops.conv1d_k3_im2col(xr.to(dev), stride, 3 * C).float().cpu(), ref)


@pytest.mark.parametrize("T,masked", [(49, False), (200, True), (333, False)])
def test_attention_fwd_gated_relative_position_bias(dev, T, masked):
    """slam_attn_fwd with rp_gate / rp_tab (WavLM) vs torch fp32: softmax(q.k * scale + gate[b,h,q] * tab[h, k - q] [+ key mask]) v;
    slam_wavlm_gate vs the same gate arithmetic in torch"""
    from slam_llm_amd import ops
    from slam_llm_amd.host_tables import wavlm_relative_buckets
    B, H, D = 2, 3, 64
    g = torch.Generator().manual_seed(T)
    x = (torch.randn(B * T, H * D, generator=g) * 0.8).to(torch.bfloat16)
    qkv = torch.randn(B * T, 3 * H * D, generator=g).to(torch.bfloat16)
    gw, gb, ga = torch.randn(8, D, generator=g) * 0.2, torch.randn(8, generator=g) * 0.2, 1 + torch.randn(H, generator=g) * 0.3
    emb = torch.randn(40, H, generator=g)
    gate = ops.wavlm_gate(x.to(dev), gw.to(dev), gb.to(dev), ga.to(dev), B, T, H)
    xh = x.float().view(B, T, H, D).permute(0, 2, 1, 3)
    gl = torch.sigmoid((xh @ gw.t() + gb).view(B, H, T, 2, 4).sum(-1))
    gate_ref = gl[..., 0] * (gl[..., 1] * ga[None, :, None] - 1.0) + 2.0
    assert torch.allclose(gate[:, :, :T].cpu(), gate_ref, atol=2e-5, rtol=1e-5)
    buckets = wavlm_relative_buckets(T, 40, 24)
    tab = ops.relpos_table(emb.index_select(0, buckets).t().contiguous().to(dev))
    qd = qkv.to(dev)
    vt = ops.head_rope_transpose(qd, 2 * H * D, B, T, H, D)
    km = None
    if masked:
        km = torch.zeros((B, vt.shape[-1]), dtype=torch.uint8)
        km[0, :T] = 1
        km[1, : T - 37] = 1
    out, _ = ops.attn_fwd(qd[:, : H * D], qd[:, H * D: 2 * H * D], vt, B, T, H, H, D, False, D ** -0.5,
                          key_mask=km.to(dev) if km is not None else None, relpos=(gate, tab, T))
    q, k, v = (qkv.float()[:, i * H * D:(i + 1) * H * D].view(B, T, H, D).transpose(1, 2) for i in range(3))
    rel = torch.arange(T)[None, :] - torch.arange(T)[:, None]
    bias = emb[buckets[rel + T - 1]].permute(2, 0, 1)                       # [H, q, k]
    sc = q @ k.transpose(2, 3) * D ** -0.5 + gate_ref[..., None] * bias[None]
    if km is not None:
        sc = sc.masked_fill(km[:, None, None, :T] == 0, float("-inf"))
    ref = (torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(B * T, H * D)
    err = (out.float().cpu() - ref).abs().max().item()
    assert err < 3e-2, err


@pytest.mark.parametrize("B,T,C", [(2, 700, 64), (3, 257, 512)])
def test_groupnorm_time_gelu_backward_matches_autograd(dev, B, T, C):
    """slam_groupnorm_time_gelu_bwd (first conv layer of the base-geometry extractors, WavLM.py:428-441 / modeling_hubert.py:153-176) vs
    torch autograd of gelu(group_norm(x, groups = C)) in fp32 on the same fp32 x and bf16 dy: dx to bf16 rounding, dgamma / dbeta to
    1e-4 relative; the accumulate form adds."""
    from slam_llm_amd import ops
    g = torch.Generator().manual_seed(B * T)
    x = torch.randn(B * T, C, generator=g) * 2.0 + 0.5
    wgt, bias = 1 + 0.2 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g)
    dy = torch.randn(B * T, C, generator=g).to(torch.bfloat16)
    xd = x.to(dev)
    y, stats = ops.groupnorm_time_gelu(xd, B, T, wgt.to(dev), bias.to(dev), 1e-5, stats=True)
    dgam, dbet = torch.full((C,), 3.0, device=dev), torch.full((C,), 3.0, device=dev)
    dx = ops.groupnorm_time_gelu_bwd(xd, stats, wgt.to(dev), bias.to(dev), dy.to(dev), B, T, dgam, dbet)
    xr, wr, br = x.clone().requires_grad_(True), wgt.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    ref = F.gelu(F.group_norm(xr.view(B, T, C).transpose(1, 2), C, wr, br, 1e-5)).transpose(1, 2).reshape(B * T, C)
    (ref * dy.float()).sum().backward()
    assert_close(y, ref.detach(), atol=2e-2, rtol=2e-2, what="gn forward")
    assert_close(dx, xr.grad, atol=4e-3 * float(xr.grad.abs().max()), rtol=1e-2, what="gn dx")
    assert torch.allclose(dgam.cpu(), wr.grad, atol=1e-3, rtol=1e-4) and torch.allclose(dbet.cpu(), br.grad, atol=1e-3, rtol=1e-4)
    ops.groupnorm_time_gelu_bwd(xd, stats, wgt.to(dev), bias.to(dev), dy.to(dev), B, T, dgam, dbet, accumulate=True)
    assert torch.allclose(dgam.cpu(), 2 * wr.grad, atol=2e-3, rtol=1e-4) and torch.allclose(dbet.cpu(), 2 * br.grad, atol=2e-3, rtol=1e-4)


def test_wavlm_gate_backward_matches_autograd(dev):
    """slam_wavlm_gate_bwd (unfrozen WavLM, modules.py:522-531): dL/d(grep_linear.weight | bias), dL/d(grep_a) and dL/d(attention input)
    against torch autograd of the same gate arithmetic on the same bf16 input, for a random dL/d(gate): cosine >= 0.9999 on the
    parameter gradients (they are sums of bf16-rounded per-frame terms, like every Linear bias gradient of the step), dx to bf16
    rounding."""
    from slam_llm_amd import ops
    B, T, H, D = 2, 333, 3, 64
    M = B * T
    g = torch.Generator().manual_seed(21)
    x = (torch.randn(M, H * D, generator=g) * 0.8).to(torch.bfloat16)
    gw, gb, ga = torch.randn(8, D, generator=g) * 0.2, torch.randn(8, generator=g) * 0.2, 1 + torch.randn(H, generator=g) * 0.3
    Tp = ops.round_up(T, 64)
    dgate = torch.zeros(B, H, Tp)
    dgate[..., :T] = torch.randn(B, H, T, generator=g)
    xd = x.to(dev)
    dv8, da_term, dx = ops.wavlm_gate_bwd(xd, gw.to(dev), gb.to(dev), ga.to(dev), dgate.to(dev), B, T, H)
    d_w = torch.zeros(8, D, device=dev)
    ops.skinny_gram(dv8, xd.view(M * H, D), d_w, D, 1)
    d_b = torch.zeros(8, device=dev)
    ops.colsum(dv8, d_b)
    tmp = torch.zeros(da_term.shape[1], device=dev)
    ops.colsum(da_term, tmp)
    xr = x.float().requires_grad_(True)
    wr, br, ar = gw.clone().requires_grad_(True), gb.clone().requires_grad_(True), ga.clone().requires_grad_(True)
    gl = torch.sigmoid((xr.view(B, T, H, D).permute(0, 2, 1, 3) @ wr.t() + br).view(B, H, T, 2, 4).sum(-1))
    gate = gl[..., 0] * (gl[..., 1] * ar[None, :, None] - 1.0) + 2.0
    (gate * dgate[..., :T]).sum().backward()
    for name, got, ref in (("grep_linear.weight", d_w, wr.grad), ("grep_linear.bias", d_b, br.grad), ("grep_a", tmp[:H], ar.grad)):
        got = got.float().cpu()
        cs = float((got * ref).sum() / (got.norm() * ref.norm()))
        G.floor_check(cs, 0.9999, f"{name}: cosine {cs}, norms {float(got.norm())} / {float(ref.norm())}")
        assert abs(float(got.norm() / ref.norm()) - 1) < 5e-3, f"{name}: cosine {cs}, norms {float(got.norm())} / {float(ref.norm())}"
    assert_close(dx, xr.grad, atol=2e-3, rtol=1e-2, what="gate dx")


def test_weight_norm_backward_and_bucket_gradient(dev):
    """slam_weight_norm_bwd: the chain rule of nn.utils.weight_norm(dim=2) (WavLM's positional conv, WavLM.py:378-386) vs torch autograd, fp32,
    plain and accumulating; slam_relpos_bucket_grad: d(relative_attention_bias.weight) gathered from the gradient of the per-distance
    table (modules.py:444-455) vs index_add_."""
    from slam_llm_amd import ops
    from slam_llm_amd.host_tables import wavlm_relative_buckets
    g = torch.Generator().manual_seed(3)
    d, gch, K = 128, 32, 16
    v = torch.randn(d, gch, K, generator=g) * 0.1
    gg = v.norm(dim=(0, 1), keepdim=True) * (1 + 0.1 * torch.randn(1, 1, K, generator=g))
    dw = torch.randn(d, gch, K, generator=g)
    vr, gr = v.clone().requires_grad_(True), gg.clone().requires_grad_(True)
    ((gr * vr / vr.norm(dim=(0, 1), keepdim=True)) * dw).sum().backward()
    dg, dv = torch.full((1, 1, K), 7.0, device=dev), torch.full((d, gch, K), 7.0, device=dev)
    ops.weight_norm_bwd(dw.to(dev), v.to(dev), gg.to(dev), dg, dv)
    assert torch.allclose(dg.cpu(), gr.grad, atol=1e-5, rtol=1e-4) and torch.allclose(dv.cpu(), vr.grad, atol=1e-5, rtol=1e-4)
    ops.weight_norm_bwd(dw.to(dev), v.to(dev), gg.to(dev), dg, dv, accumulate=True)
    assert torch.allclose(dg.cpu(), 2 * gr.grad, atol=2e-5, rtol=1e-4) and torch.allclose(dv.cpu(), 2 * vr.grad, atol=2e-5, rtol=1e-4)
    T, H, nb = 200, 3, 40
    buckets = wavlm_relative_buckets(T, nb, 24)
    d_vals = torch.randn(H, 2 * T - 1, generator=g)
    d_tab = ops.relpos_table(d_vals.to(dev))
    out = torch.full((nb, H), 5.0, device=dev)
    ops.relpos_bucket_grad(d_tab, buckets.to(torch.int32).to(dev), nb, out)
    ref = torch.zeros(nb, H).index_add_(0, buckets, d_vals.t().contiguous())
    assert torch.allclose(out.cpu(), ref, atol=1e-4, rtol=1e-5)
    ops.relpos_bucket_grad(d_tab, buckets.to(torch.int32).to(dev), nb, out, accumulate=True)
    assert torch.allclose(out.cpu(), 2 * ref, atol=2e-4, rtol=1e-5)


@pytest.mark.parametrize("B,Tq,Tk,H,masked", [(2, 8, 8, 3, False), (2, 64, 150, 2, True), (1, 100, 100, 2, False)])
def test_attention_probability_dropout_fwd_bwd(dev, B, Tq, Tk, H, masked):
    """slam_attn_fwd / slam_attn_bwd with drop_p > 0 (Q-Former self- and cross-attention shapes, D = 64, bidirectional) vs torch
    autograd through softmax(q.k) * mask / (1-p) @ v with the SAME mask, rebuilt on the host from the seed (tests/golden_util.py)"""
    from slam_llm_amd import ops
    D, pdrop, seed = 64, 0.1, 0x1234567
    g = torch.Generator().manual_seed(Tq * 31 + Tk)
    q = torch.randn(B * Tq, H * D, generator=g).to(torch.bfloat16)
    kv = torch.randn(B * Tk, 2 * H * D, generator=g).to(torch.bfloat16)
    do = torch.randn(B * Tq, H * D, generator=g).to(torch.bfloat16)
    qd, kvd, dod = q.to(dev), kv.to(dev), do.to(dev)
    vt = ops.head_rope_transpose(kvd, H * D, B, Tk, H, D)
    kt = ops.head_rope_transpose(kvd, 0, B, Tk, H, D)
    qt = ops.head_rope_transpose(qd, 0, B, Tq, H, D)
    dot = ops.head_rope_transpose(dod, 0, B, Tq, H, D)
    Tqp, Tkp = qt.shape[-1], kt.shape[-1]
    km = None
    if masked:
        km = torch.zeros((B, Tkp), dtype=torch.uint8)
        km[0, :Tk] = 1
        km[1, : Tk - 37] = 1
    scale = D ** -0.5
    o, lse = ops.attn_fwd(qd, kvd[:, : H * D], vt, B, Tq, H, H, D, False, scale, key_mask=km.to(dev) if masked else None, Tk=Tk,
                          drop=(pdrop, seed))
    dq = torch.empty_like(qd)
    dkv = torch.empty_like(kvd)
    ops.attn_bwd(qd, kvd[:, : H * D], kvd[:, H * D:], o, dod, lse, dq, dkv[:, : H * D], dkv[:, H * D:], B, Tq, H, H, D,
                 False, scale, key_mask=km.to(dev) if masked else None, Tk=Tk, drop=(pdrop, seed))
    keep = torch.from_numpy(G.attn_keep_mask(seed, pdrop, B, H, Tq, Tk, Tqp, Tkp))
    assert abs(float(keep.mean()) - (1 - pdrop)) < 0.03
    qf = q.float().view(B, Tq, H, D).transpose(1, 2).requires_grad_(True)
    kf = kv.float()[:, : H * D].reshape(B, Tk, H, D).transpose(1, 2).requires_grad_(True)
    vf = kv.float()[:, H * D:].reshape(B, Tk, H, D).transpose(1, 2).requires_grad_(True)
    sc = qf @ kf.transpose(2, 3) * scale
    if masked:
        sc = sc.masked_fill(km[:, None, None, :Tk] == 0, float("-inf"))
    ref = ((torch.softmax(sc, -1) * keep / (1 - pdrop)) @ vf).transpose(1, 2).reshape(B * Tq, H * D)
    ref.backward(do.float())
    assert_close(o, ref.detach(), atol=3e-2, rtol=2e-2, what="attn fwd with dropout")
    for got, want, nm in ((dq.view(B, Tq, H, D).transpose(1, 2), qf.grad, "dq"), (dkv[:, : H * D].reshape(B, Tk, H, D).transpose(1, 2), kf.grad, "dk"),
                          (dkv[:, H * D:].reshape(B, Tk, H, D).transpose(1, 2), vf.grad, "dv")):
        err = (got.float().cpu() - want).abs().max().item()
        assert err < 4e-2 * max(1.0, want.abs().max().item()), (nm, err, want.abs().max().item())
    # a different seed gives a different mask; p = 0 equals the plain kernel
    o2, _ = ops.attn_fwd(qd, kvd[:, : H * D], vt, B, Tq, H, H, D, False, scale, key_mask=km.to(dev) if masked else None, Tk=Tk,
                         drop=(pdrop, seed + 1))
    assert not torch.equal(o2, o)
