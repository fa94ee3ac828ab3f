"""Per-kernel table of rocprofv3 --pmc passes: python tools/pmc_table.py <out.md> <title> <dir> [<dir> ...]
Every csv under the directories is read; counters are averaged per dispatch and per (kernel, grid size) -- so different shapes of one kernel
stay apart; the kernel duration comes from the Start/End timestamps of the same rows when the csv carries them."""
import collections
import csv
import glob
import re
import sys


def short(name):
    m = re.search(r"::([A-Za-z0-9_]+(<[^(]*>)?)\(", name)
    n = m.group(1) if m else name.split("(")[0]
    return n.replace(" ", "")


def main():
    out, title, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            seen = set()
            for row in csv.DictReader(open(f)):
                k = (short(row["Kernel_Name"]), int(row.get("Grid_Size", 0) or 0))
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
                did = row.get("Dispatch_Id")
                if did not in seen and row.get("Start_Timestamp") and row.get("End_Timestamp"):
                    seen.add(did)
                    dur[(k, d)].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    counters = sorted({c for v in acc.values() for c in v})
    with open(out, "w") as fh:
        fh.write(f"# {title}\n\nAverages per dispatch (first launch of each shape dropped where more than two were profiled); durations in us from the pass's own "
                 "timestamps (profiled passes run ~2-3 % slower than un-profiled ones).\n\n")
        fh.write("| kernel | grid | n | us | " + " | ".join(counters) + " |\n|---|---|---|---|" + "---|" * len(counters) + "\n")
        for k in sorted(acc, key=lambda k: (-max(len(v) for v in acc[k].values()), k)):
            if not any(s in k[0] for s in ("gemm", "attn", "logmel", "lora_hop")):
                continue
            ds = [x for (kk, d), v in dur.items() if kk == k for x in (v[1:] if len(v) > 2 else v)]
            cells = []
            for c in counters:
                v = acc[k].get(c, [])
                v = v[1:] if len(v) > 2 else v
                cells.append(f"{sum(v) / len(v):.4g}" if v else "")
            nd = max(len(v) for v in acc[k].values())
            fh.write(f"| `{k[0][:70]}` | {k[1]} | {nd} | {(sum(ds) / len(ds) if ds else float('nan')):.1f} | " + " | ".join(cells) + " |\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
