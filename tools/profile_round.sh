#!/bin/bash
# GPU box: bench lines for the non-headline workloads + rocprofv3 kernel-trace summaries + PMC traffic passes (run from the repo root)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02
mkdir -p $O
export TMPDIR=/tmp
cd $R
for wl in c1 c2 c4; do
  python bench.py --workload $wl --steps 8 --warmup 3 > $O/bench_$wl.json 2> $O/bench_$wl.err || echo "bench $wl failed"
done
(cd /tmp && rocprofv3 --kernel-trace --stats -d $O/prof_c3 -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/prof_c3.json 2> $O/prof_c3.err)
python tools/rocpd_stats.py $(ls $O/prof_c3/*/*.db | head -1) $O/r02_c3_kernel_stats.md > /dev/null
(cd /tmp && rocprofv3 --kernel-trace --stats -d $O/prof_c4 -- python $R/bench.py --workload c4 --steps 4 --warmup 2 --no-cpu-baseline > $O/prof_c4.json 2> $O/prof_c4.err)
python tools/rocpd_stats.py $(ls $O/prof_c4/*/*.db | head -1) $O/r02_c4_kernel_stats.md > /dev/null
(cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.json 2> $O/pmc_fetch.err)
(cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_write.json 2> $O/pmc_write.err)
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/traffic.json "python bench.py --steps 2 --warmup 1 --no-cpu-baseline" > $O/traffic.txt 2>&1
rm -rf $O/prof_c3/*/*.db $O/prof_c4/*/*.db $O/pmc_fetch $O/pmc_write   # keep the merged output small
ls -la $O
tail -3 $O/traffic.txt
for wl in c1 c2 c4; do python - <<PY
import json
try:
    d=json.load(open("$O/bench_$wl.json"))
    print("$wl", round(d["ms_per_step"],2), "ms", round(d["value"],1), "audio-s/s mfu", round(d["mfu"],3), "cpu", d.get("cpu_baseline",{}).get("value"))
except Exception as e: print("$wl", "ERR", e, open("$O/bench_$wl.err").read()[-800:])
PY
done
