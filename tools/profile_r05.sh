#!/bin/bash
# GPU box, round 5 (HEAD): rocprofv3 kernel-trace summary of the C3 bench, PMC traffic passes (FETCH_SIZE / WRITE_SIZE, separate runs),
# bench lines of every workload, the ragged rows incl. the 288 GB-sized one.  Run from the repo root; everything lands under gpurun_out/r05/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05
mkdir -p $O
export TMPDIR=/tmp
cd $R
(cd /tmp && rocprofv3 --kernel-trace --stats -d $O/prof_c3 -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/prof_c3.json 2> $O/prof_c3.err)
python tools/rocpd_stats.py $(ls $O/prof_c3/*/*.db | head -1) $O/r05_c3_kernel_stats.md > /dev/null
if [ "${1:-all}" = "all" ]; then
  (cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.json 2> $O/pmc_fetch.err)
  (cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_write.json 2> $O/pmc_write.err)
  python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/traffic.json "python bench.py --steps 2 --warmup 1 --no-cpu-baseline" > $O/traffic.txt 2>&1
  python bench.py --steps 8 --warmup 3 > $O/bench_c3.json 2> $O/bench_c3.err || echo "bench c3 failed"
  for wl in c1 c2 c4; do
    python bench.py --workload $wl --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_$wl.json 2> $O/bench_$wl.err || echo "bench $wl failed"
  done
  python tools/ragged_bench.py padded,ragged,ragged_sum_hbm,c5 > $O/ragged.txt 2> $O/ragged.err || echo "ragged bench failed"
fi
rm -rf $O/prof_c3/*/*.db $O/pmc_fetch $O/pmc_write
ls -la $O
head -40 $O/r05_c3_kernel_stats.md
