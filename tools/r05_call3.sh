#!/bin/bash
# round 5, GPU call 3: the 288 GB-sized ragged row, a per-call kernel trace of the C3 step, the suite's cosine margins
O=gpurun_out/r05c
mkdir -p $O
timeout 900 python tools/ragged_bench.py ragged,ragged_sum_hbm > $O/ragged.txt 2> $O/ragged.err || tail -5 $O/ragged.err
grep -v "^{" $O/ragged.txt | tail -4
R=$PWD
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_c3 -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $R/$O/prof_c3.json 2> $R/$O/prof_c3.err)
ls $O/prof_c3/* | head
python - <<'PY'
import csv, glob, collections, os
O = "gpurun_out/r05c"
f = glob.glob(O + "/prof_c3/**/*kernel_trace.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    # keep only what the analysis needs: name (trimmed), start, end, grid -- and drop the raw trace (tens of MB)
    with open(O + "/c3_trace_small.tsv", "w") as out:
        for r in rows:
            out.write(f"{r['Kernel_Name'][:90]}\t{r['Start_Timestamp']}\t{r['End_Timestamp']}\t{r.get('Grid_Size','')}\t{r.get('Workgroup_Size','')}\n")
    os.remove(f[0])
PY
ls -la $O
SLAM_TEST_MARGINS=$O/margins.tsv timeout 1200 python -m pytest tests -m gpu -q -k "not full_depth" -p no:cacheprovider > $O/gpu_suite.log 2>&1
tail -3 $O/gpu_suite.log
python tools/margins_report.py $O/margins.tsv > $O/margins.md; head -30 $O/margins.md
