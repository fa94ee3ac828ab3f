#!/bin/bash
# round 5, GPU call 16: bench.py with per-launch events on every 4th timed step: default invocation, the driver's (20 steps), and the
# same under rocprofv3 (its dominant-kernel average must agree with the sampled HIP-event average)
O=gpurun_out/r05p
mkdir -p $O
R=$PWD
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err || tail -5 $O/bench_default.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20.json 2> $O/bench_20.err || tail -5 $O/bench_20.err
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $R/$O/bench_prof.json 2> $R/$O/bench_prof.err)
python tools/rocpd_stats.py $(ls $O/prof/*/*.db | head -1) $O/kernel_stats.md > /dev/null 2>&1
rm -rf $O/prof
python - <<'PY'
import json
for k in ("default","20","prof"):
    d=json.load(open(f"gpurun_out/r05p/bench_{k}.json")); r=d["roofline"]
    print(k, "ms/step", round(d["ms_per_step"],2), "value", round(d["value"],1), "frac", round(r["frac"],4), "avg_launch_ms", round(r["avg_launch_ms"],4), "launches/step", r["launches_per_step"], "|", r["timing"][:60], "| cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
head -6 $O/kernel_stats.md | tail -2 | cut -c1-200
