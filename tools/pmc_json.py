"""profiles/pmc.json from two rocprofv3 --pmc passes over tools/pmc_gemm.py (and, optionally, an attention command):
    python tools/pmc_json.py <out.json> <source text> <pass-1 dir> <pass-2 dir> [<pass-1 dir> <pass-2 dir> ...]
pass 1: SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
pass 2: SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
Derived per (kernel, grid): duration (the pass's own timestamps), effective clock = GRBM_GUI_ACTIVE / 8 XCDs / duration, mfma_util =
SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs), shares of SQ_WAVE_CYCLES (parked = SQ_WAIT_ANY, issuing =
SQ_ACTIVE_INST_ANY, issue-stalled = the rest), L2 hit rate = TCC_HIT / (TCC_HIT + TCC_MISS).  The schema bench.py's pmc_for() reads."""
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    m = re.search(r"::([A-Za-z0-9_]+(<[^(]*>)?)\(", name)
    n = m.group(1) if m else name.split("(")[0]
    return n.replace(" ", "")


def read(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        seen = set()
        for row in csv.DictReader(open(f)):
            k = (short(row["Kernel_Name"]), int(row.get("Grid_Size", 0) or 0))
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
            did = row.get("Dispatch_Id")
            if did not in seen and row.get("Start_Timestamp") and row.get("End_Timestamp"):
                seen.add(did)
                dur[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    return acc, dur


def mean(v):
    v = v[1:] if len(v) > 2 else v      # first launch of a shape dropped
    return sum(v) / len(v) if v else 0.0


def main():
    out, source, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    kernels = {}
    for p1, p2 in zip(dirs[0::2], dirs[1::2]):
        a1, d1 = read(p1)
        a2, _ = read(p2)
        for k in a1:
            if not any(s in k[0] for s in ("gemm_nt", "attn_")) or k not in d1:
                continue
            c = {n: mean(v) for n, v in a1[k].items()}
            c2 = {n: mean(v) for n, v in a2.get(k, {}).items()}
            us = mean(d1[k])
            gui = c.get("GRBM_GUI_ACTIVE", 0.0)
            if us <= 0 or gui <= 0 or us < 20:
                continue
            wc = max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0)
            parked, issuing = c.get("SQ_WAIT_ANY", 0.0) / wc, c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
            hit, miss = c2.get("TCC_HIT_sum", 0.0), c2.get("TCC_MISS_sum", 0.0)
            kernels[f"{k[0]} @ grid {k[1]}"] = dict(
                duration_us=round(us, 1), effective_clock_GHz=round(gui / 8 / us / 1e3, 3),
                mfma_util=round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8 * 256 * 4), 3),
                wave_cycles_share=dict(parked_waitcnt_or_barrier=round(parked, 3), issue_stalled=round(max(0.0, 1 - parked - issuing), 3), issuing=round(issuing, 3)),
                avg_waves_per_simd=round(c.get("SQ_WAVE_CYCLES", 0.0) * 4 / (gui / 8 * 256 * 4), 2),
                l2_hit_rate=round(hit / (hit + miss), 3) if hit + miss > 0 else None,
                lds_bank_conflict_cycles=c2.get("SQ_LDS_BANK_CONFLICT", 0.0),
                insts=dict(mfma=c2.get("SQ_INSTS_MFMA", 0.0), valu_incl_mfma=c2.get("SQ_INSTS_VALU", 0.0), lds=c2.get("SQ_INSTS_LDS", 0.0),
                           vmem=c2.get("SQ_INSTS_VMEM", 0.0), salu=c2.get("SQ_INSTS_SALU", 0.0)),
                launches_profiled=len(d1[k]))
    json.dump(dict(source=source, units="GRBM_GUI_ACTIVE is summed over the 8 XCDs: effective clock = GRBM_GUI_ACTIVE / 8 / kernel duration; SQ_VALU_MFMA_BUSY_CYCLES "
                   "counts cycles summed over SIMDs: mfma_util = MFMA_BUSY / (GRBM_GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs); SQ_WAVE_CYCLES / SQ_WAIT_* / "
                   "SQ_ACTIVE_INST_* count quad-cycles", kernels=kernels), open(out, "w"), indent=1)
    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["duration_us"]):
        print(f"{k[:70]:70s} {v['duration_us']:8.1f} us  {v['effective_clock_GHz']:.3f} GHz  mfma {v['mfma_util']:.3f}  parked {v['wave_cycles_share']['parked_waitcnt_or_barrier']:.3f}  L2 {v['l2_hit_rate']}")


if __name__ == "__main__":
    main()
