"""CPU: invariants of the compiled gfx950 code that the hand-scheduled kernels rely on.

The k-loop of gemm_nt_w4_kernel, the tile loops of attn_fwd_kernel, attn_bwd_dkdv_ring_kernel and attn_bwd_dq_ring_kernel place their LDS reads, LDS-DMA and
MFMAs as `asm volatile` statements whose result registers are "ready" for the compiler at once.  If hipcc ever SPILLS such a
register (stores it to scratch before the data has landed) the kernel computes garbage -- it happened once with three
instantiations of the GEMM tile body (876 bytes of scratch, wrong results).  Zero scratch is therefore a build invariant of the
attention kernels, the pipelined GEMM and the whole-tile instantiation (SK = 0) of the 4-wave GEMM.  Its split-K instantiations (512
registers, slab / ticket code around the loop) are allowed to park loop-INVARIANT values in scratch outside the hand-ordered region: the
invariant there is that no scratch STORE appears between the first inline-asm LDS read and the last inline-asm MFMA (prologue reads +
the whole k-loop)."""
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


def _scratch_by_kernel(src, extra):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *extra, "-S", "--cuda-device-only",
                        os.path.join(CSRC, src), "-o", out], check=True, capture_output=True, cwd=CSRC)
        text = open(out).read()
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
