"""Sum rocprofv3 --pmc csv counters per kernel name: python tools/pmc_sum.py <dir> [substring]"""
import csv, glob, sys, collections
d, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if sub in k:
            acc[k[:60]][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[(k[:60], row["Counter_Name"])] += 1
for k, v in acc.items():
    print(k)
    for c, x in sorted(v.items()):
        print(f"   {c:28s} {x:16.0f}  (per dispatch {x / cnt[(k, c)]:14.0f})")
