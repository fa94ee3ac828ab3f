"""CPU: invariants of the compiled gfx950 code that the hand-scheduled kernels rely on.

The k-loop of gemm_nt_w4_kernel, the tile loops of attn_fwd_kernel, attn_bwd_dkdv_ring_kernel and attn_bwd_dq_ring_kernel place their LDS reads, LDS-DMA and
MFMAs as `asm volatile` statements whose result registers are "ready" for the compiler at once.  If hipcc ever SPILLS such a
register (stores it to scratch before the data has landed) the kernel computes garbage -- it happened once with three
instantiations of the GEMM tile body (876 bytes of scratch, wrong results).  Zero scratch is therefore a build invariant of the
attention kernels, the pipelined GEMM and the whole-tile instantiation (SK = 0) of the 4-wave GEMM.  Its split-K instantiations (512
registers, slab / ticket code around the loop) are allowed to park loop-INVARIANT values in scratch outside the hand-ordered region: the
invariant there is that no scratch STORE appears between the first inline-asm LDS read and the last inline-asm MFMA (prologue reads +
the whole k-loop)."""
import functools
import os
import re
import subprocess
import tempfile
from concurrent.futures import ThreadPoolExecutor

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "slam_llm_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
# (gemm_nt_persist2_kernel is compiler-scheduled -- no asm loads -- and spills three loop-invariant dwords around the epilogue inside
# its tile loop: harmless, not part of the invariant)
REGION_RULE = ("gemm_nt_w4_kernel",)   # kernels checked by region instead of by total scratch size
FILES = {"gemm_bf16.hip": ([], ("gemm_nt_w4_kernel", "gemm_nt_pipe_kernel")),
         "attention.hip": (["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-slp-vectorize"], ("attn_fwd_kernel", "attn_bwd_dkdv_ring_kernel", "attn_bwd_dq_ring_kernel",
                                                                  "attn_bwd_dkdv_tr_kernel", "attn_bwd_dq_tr_kernel"))}


@functools.lru_cache(maxsize=None)
def _isa_text(src, extra):
    """gfx950 assembly of one translation unit, built with the Makefile's flags (cached: both tests below read it)"""
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *extra, "-S", "--cuda-device-only",
                        os.path.join(CSRC, src), "-o", out], check=True, capture_output=True, cwd=CSRC)
        return open(out).read()


def _scratch_by_kernel(src, extra):
    text = _isa_text(src, tuple(extra))
    res = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
        res[m.group(1)] = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", m.group(2)).group(1))
    for name in list(res):
        if res[name] and any(r in name for r in REGION_RULE) and _w4_split_k_form(name) != 0:
            res[name] = _scratch_in_asm_region(text, name)
    return res


def _w4_split_k_form(mangled):
    """last template argument (SK) of a gemm_nt_w4_kernel instantiation: 0 = the whole-tile kernel the headline number is quoted on --
    it must not touch scratch AT ALL (a spilled asm result in its prologue / epilogue, outside the region rule, once produced wrong
    results); 1 / 2 = the split-K instantiations, which park loop invariants in scratch outside the hand-ordered region"""
    m = re.search(r"gemm_nt_w4_kernelILi\d+ELi\d+ELb[01]ELi\d+ELb[01]ELi(\d+)EE", mangled)
    assert m, f"cannot parse the template arguments of {mangled}"
    return int(m.group(1))


def _scratch_in_asm_region(text, name):
    """number of scratch STORES between the first inline-asm ds_read and the last inline-asm MFMA of kernel `name` (a spilled asm
    result is a store inside that region; reloading a loop-invariant value there is harmless)"""
    start = text.index("\n" + name + ":")
    lines = text[start: text.index(".end_amdhsa_kernel", start)].split("\n")
    inasm, first, last = False, None, None
    for i, l in enumerate(lines):
        if "ASMSTART" in l:
            inasm = True
        elif "ASMEND" in l:
            inasm = False
        elif inasm and "ds_read_b128" in l and first is None:
            first = i
        elif inasm and "v_mfma" in l:
            last = i
    assert first is not None and last is not None and last > first, f"{name}: hand-ordered region not found"
    return sum(1 for l in lines[first:last + 1] if "scratch_store" in l)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_hand_scheduled_kernels_use_no_scratch():
    with ThreadPoolExecutor(3) as ex:
        futs = {src: ex.submit(_scratch_by_kernel, src, extra) for src, (extra, _) in FILES.items()}
    for src, (_, names) in FILES.items():
        scratch = futs[src].result()
        for name in names:
            hits = {k: v for k, v in scratch.items() if name in k}
            assert hits, f"{name}: no instance found in {src}"
            assert all(v == 0 for v in hits.values()), f"{src}: scratch in {', '.join(k for k, v in hits.items() if v)}"


# ---- hand-placed LDS reads: every use sits behind a wait that covers the read -------------------------------------------------------
# The attention tile loops request their LDS fragments with `asm volatile("ds_read_b128 / ds_read_b64_tr_b16 ...")` and wait with COUNTED
# `s_waitcnt lgkmcnt(N)` statements (attn_common.h: lds_read128 / lds_read_tr / lds_wait).  To the compiler the result registers are ready at once, so a
# count that does not cover a read is a silent read-before-landed (there is no hardware interlock on an outstanding LDS return).  Round 6 found one such
# count in the D = 64 dK / dV kernel (a top-up fragment requested behind a younger one): the shipped binary was right only because hipcc had sunk the two
# MFMAs below the next full wait.  This test reads the ISA instead of trusting the source: a linear scan per kernel, LDS operations return in order,
# SMEM out of order (while one is pending only lgkmcnt(0) lands anything); pending reads are forgotten at labels and branches (control flow is not followed).
_VREG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def _vregs(text):
    out = set()
    for m in _VREG.finditer(text):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def _unwaited_lds_uses(lines):
    pending, issued, smem_pending, bad = [], 0, False, []     # pending: (sequence number, destination registers, text), oldest first
    for ln in lines:
        t = ln.split(";")[0].strip()
        if not t or t.startswith("."):
            if re.match(r"^\S+:$", t):
                pending, issued, smem_pending = [], 0, False
            continue
        op = t.split()[0]
        if op.startswith("s_cbranch") or op in ("s_branch", "s_endpgm", "s_setpc_b64", "s_swappc_b64"):
            pending, issued, smem_pending = [], 0, False
            continue
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", t)
            if m or re.fullmatch(r"s_waitcnt\s+0", t):
                n = int(m.group(1)) if m else 0
                if n == 0:
                    pending, smem_pending = [], False
                elif not smem_pending:
                    pending = [q for q in pending if q[0] > issued - n]
            continue
        body = t[len(op):]
        if op.startswith("ds_"):
            ops_ = [x.strip() for x in body.split(",")]
            is_read = any(k in op for k in ("read", "permute", "swizzle", "consume", "append", "_rtn", "load"))
            dest = _vregs(ops_[0]) if is_read else set()
            srcs = _vregs(",".join(ops_[1:] if is_read else ops_))
            bad += [(t, q[2]) for q in pending if srcs & q[1]]
            issued += 1
            if is_read:
                pending = [q for q in pending if not (q[1] & dest)] + [(issued, dest, t)]
            continue
        if op.startswith(("s_load", "s_buffer_load")) or op in ("s_memtime", "s_memrealtime", "s_sendmsg", "s_sendmsghalt"):
            issued += 1
            smem_pending = True
            continue
        used = _vregs(body)
        bad += [(t, q[2]) for q in pending if used & q[1]]
    return bad


def test_lds_wait_checker_catches_an_uncovered_read():
    """the checker itself: the round-6 pattern (fragment 2 requested behind fragment 3, waited for with a count that covers neither) is flagged when the
    MFMA is NOT sunk below the full wait, and passes in the order hipcc happened to emit"""
    req = ["ds_read_b64_tr_b16 v[62:63], v76 offset:0x1000", "ds_read_b64_tr_b16 v[64:65], v76 offset:0x1200",       # fragment 3
           "ds_read_b64_tr_b16 v[50:51], v56 offset:0x1000", "ds_read_b64_tr_b16 v[52:53], v56 offset:0x1200"]       # fragment 2, requested later
    mf2, mf3 = "v_mfma_f32_16x16x32_bf16 v[22:25], v[50:53], v[66:69], v[22:25]", "v_mfma_f32_16x16x32_bf16 v[18:21], v[62:65], v[66:69], v[18:21]"
    hazard = req + ["s_waitcnt lgkmcnt(4)", mf2, "s_waitcnt lgkmcnt(0)", mf3]
    sunk = req + ["s_waitcnt lgkmcnt(4)", "s_waitcnt lgkmcnt(0)", mf2, mf3]
    assert {u for u, _ in _unwaited_lds_uses(hazard)} == {mf2}          # (both halves of fragment 2 are still in flight)
    assert _unwaited_lds_uses(sunk) == []
    assert _unwaited_lds_uses(req + ["s_waitcnt lgkmcnt(2)", mf3, "s_waitcnt lgkmcnt(0)", mf2]) == []      # in-order return: the older fragment has landed at count 2
    assert len(_unwaited_lds_uses(req[:2] + ["s_load_dword s4, s[0:1], 0x0", "s_waitcnt lgkmcnt(1)", mf3])) == 2   # SMEM pending: a non-zero count proves nothing


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_hand_placed_lds_reads_are_covered_by_their_waits():
    with ThreadPoolExecutor(3) as ex:      # (cached when the scratch test ran first)
        texts = {src: ex.submit(_isa_text, src, tuple(extra)) for src, (extra, _) in FILES.items()}
    for src, need in (("attention.hip", 40), ("gemm_bf16.hip", 10)):      # (every kernel of both translation units, hand-ordered or not)
        checked = 0
        for m in re.finditer(r"\n(_Z\S+):\s*;[^\n]*\n(.*?)\n\s*s_endpgm", texts[src].result(), re.S):
            bad = _unwaited_lds_uses(m.group(2).split("\n"))
            assert not bad, f"{m.group(1)}: {len(bad)} use(s) of a register an LDS read is still filling, e.g. `{bad[0][0]}` after `{bad[0][1]}`"
            checked += 1
        assert checked >= need, f"only {checked} kernels found in the ISA of {src}"
