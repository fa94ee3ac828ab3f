"""GPU box (tools): which kernels that are NOT this library's run inside one steady-state C3 step (torch fills / copies / elementwise), from a rocprofv3 kernel-trace CSV.
python tools/step_torch_kernels.py <dir with *_kernel_trace.csv>"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
marks = [i for i, n in enumerate(names) if "logmel_init_kernel" in n]       # first kernel of every step
lo, hi = marks[-2], marks[-1]                                                  # the last complete step
acc = collections.defaultdict(lambda: [0, 0.0])
tot = 0.0
for r in rows[lo:hi]:
    n = r["Kernel_Name"]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    if "anonymous namespace" in n and "at::native" not in n:
        continue
    acc[n[:150]][0] += 1
    acc[n[:150]][1] += d
span = (int(rows[hi]["Start_Timestamp"]) - int(rows[lo]["Start_Timestamp"])) / 1e6
print(f"step: {hi - lo} dispatches, kernel time {tot / 1e3:.2f} ms, span {span:.2f} ms, idle {span - tot / 1e3:.2f} ms")
for n, (c, d) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{c:5d} {d:9.1f} us  {n}")
