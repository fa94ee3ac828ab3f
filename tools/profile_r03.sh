#!/bin/bash
# GPU box, round 3: rocprofv3 kernel-trace summary of the C3 bench, PMC traffic passes (FETCH_SIZE / WRITE_SIZE, separate runs),
# attention fabric traffic with the hardware vs the XCD-aware workgroup order, bench lines of the other workloads.
# Run from the repo root; everything lands under gpurun_out/r03/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03
mkdir -p $O
export TMPDIR=/tmp
cd $R
(cd /tmp && rocprofv3 --kernel-trace --stats -d $O/prof_c3 -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/prof_c3.json 2> $O/prof_c3.err)
python tools/rocpd_stats.py $(ls $O/prof_c3/*/*.db | head -1) $O/r03_c3_kernel_stats.md > /dev/null
(cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.json 2> $O/pmc_fetch.err)
(cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_write.json 2> $O/pmc_write.err)
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/traffic.json "python bench.py --steps 2 --warmup 1 --no-cpu-baseline" > $O/traffic.txt 2>&1
for order in hw xcd; do
  (cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_attn_$order -- python $R/tools/attn_xcd_ab.py --only $order > /dev/null 2> $O/pmc_attn_$order.err)
  (cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_attnw_$order -- python $R/tools/attn_xcd_ab.py --only $order > /dev/null 2>> $O/pmc_attn_$order.err)
  python tools/pmc_traffic.py $O/pmc_attn_$order $O/pmc_attnw_$order $O/traffic_attn_$order.json "python tools/attn_xcd_ab.py --only $order" > $O/traffic_attn_$order.txt 2>&1
done
for wl in c1 c2 c4; do
  python bench.py --workload $wl --steps 8 --warmup 3 > $O/bench_$wl.json 2> $O/bench_$wl.err || echo "bench $wl failed"
done
rm -rf $O/prof_c3/*/*.db $O/pmc_fetch $O/pmc_write $O/pmc_attn_hw $O/pmc_attn_xcd $O/pmc_attnw_hw $O/pmc_attnw_xcd
ls -la $O
tail -4 $O/traffic.txt $O/traffic_attn_hw.txt $O/traffic_attn_xcd.txt
