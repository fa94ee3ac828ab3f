"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, --output-format csv) of `bench.py` into
profiles/traffic.json: HBM-side bytes per launch of every product kernel, as bench.py's `roofline.traffic` reads them.

    python tools/pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json> "<command profiled>"

Units / corrections (/opt/skills/guides/MI355X_MICROARCH.md, HBM section): both counters are reported in KiB; on gfx950 FETCH_SIZE
counts 128-byte requests at 64 bytes, i.e. exactly half of a wide coalesced streaming read -> doubled here.  WRITE_SIZE is taken as
reported (uncalibrated in the guide; for the GEMM it matches the C matrix to 2 %).  Infinity-Cache hits are included in both
(they are fabric-side requests of the L2), so `traffic` is an upper bound on true HBM bytes."""
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


def collect(d, counter):
    tot, cnt = collections.Counter(), collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            k = short(row["Kernel_Name"])
            tot[k] += float(row["Counter_Value"])
            cnt[k] += 1
    return tot, cnt


def main():
    dfetch, dwrite, out = sys.argv[1], sys.argv[2], sys.argv[3]
    cmd = sys.argv[4] if len(sys.argv) > 4 else ""
    ft, fc = collect(dfetch, "FETCH_SIZE")
    wt, wc = collect(dwrite, "WRITE_SIZE")
    kernels = {}
    for k in ft:
        if fc[k] == 0 or wc.get(k, 0) == 0:
            continue
        kernels[k] = dict(fetch_bytes_per_launch=2.0 * 1024.0 * ft[k] / fc[k], write_bytes_per_launch=1024.0 * wt[k] / wc[k],
                          launches_profiled=int(fc[k]))
    res = dict(source=f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `{cmd}`; FETCH_SIZE x2 (gfx950 correction), "
                      "KiB -> bytes; averages over every launch of the kernel in the profiled run", kernels=kernels)
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["fetch_bytes_per_launch"] * kv[1]["launches_profiled"])[:12]:
        print(f"{k:60s} n={v['launches_profiled']:5d} fetch {v['fetch_bytes_per_launch'] / 1e6:9.1f} MB  write {v['write_bytes_per_launch'] / 1e6:9.1f} MB")


if __name__ == "__main__":
    main()
