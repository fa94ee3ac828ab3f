#!/bin/bash
# GPU box, round 6 (HEAD): rocprofv3 kernel table of the C3 bench, the step cut by PHASE (roctx ranges), PMC traffic passes (FETCH_SIZE / WRITE_SIZE,
# separate runs), PMC utilisation passes on the GEMM and attention kernels (profiles/pmc.json refreshed), bench lines of every workload,
# the GEMM ceiling yardstick.  Run from the repo root; everything lands under gpurun_out/r06/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
export TMPDIR=/tmp
cd $R
(cd /tmp && rocprofv3 --kernel-trace --stats -d $O/prof_c3 -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_c3_under_rocprof.json 2> $O/prof_c3.err)
python tools/rocpd_stats.py $(ls $O/prof_c3/*/*.db | head -1) $O/r06_c3_kernel_stats.md > /dev/null
# the step by phase: roctx ranges (SLAM_ROCTX=1) joined to the kernel dispatches through the HIP runtime trace
(cd /tmp && SLAM_ROCTX=1 rocprofv3 --marker-trace --hip-runtime-trace --kernel-trace --output-format csv -d $O/prof_phase -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c3_roctx.json 2> $O/prof_phase.err)
python tools/phase_table.py $O/prof_phase $O/r06_c3_phases.md "C3 step by phase at HEAD (round 6)" > /dev/null 2> $O/phase_table.err || tail -3 $O/phase_table.err
if [ "${1:-all}" = "all" ]; then
  (cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.json 2> $O/pmc_fetch.err)
  (cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_write.json 2> $O/pmc_write.err)
  python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/traffic.json "python bench.py --steps 2 --warmup 1 --no-cpu-baseline" > $O/traffic.txt 2>&1
  P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
  P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"
  for what in gemm attn; do
    if [ $what = gemm ]; then CMD="python $R/tools/pmc_gemm.py 5"; else CMD="python $R/tools/pmc_attn_r06.py 4"; fi
    (cd /tmp && rocprofv3 --pmc $P1 --output-format csv -d $O/pmc1_$what -- $CMD > /dev/null 2> $O/pmc1_$what.err)
    (cd /tmp && rocprofv3 --pmc $P2 --output-format csv -d $O/pmc2_$what -- $CMD > /dev/null 2> $O/pmc2_$what.err)
  done
  python tools/pmc_json.py $O/pmc.json "rocprofv3 --pmc, two passes per command (counter lists in tools/profile_r06.sh) over python tools/pmc_gemm.py 5 and python tools/pmc_attn_r06.py 4 at HEAD, round 6" $O/pmc1_gemm $O/pmc2_gemm $O/pmc1_attn $O/pmc2_attn > $O/pmc_summary.txt 2> $O/pmc_json.err || tail -3 $O/pmc_json.err
  python tools/pmc_table.py $O/r06_pmc_gemm.md "GEMM, HEAD (auto rule), round 6" $O/pmc1_gemm $O/pmc2_gemm > /dev/null 2>> $O/table.err
  python tools/pmc_table.py $O/r06_pmc_attn.md "attention kernels, HEAD, round 6" $O/pmc1_attn $O/pmc2_attn > /dev/null 2>> $O/table.err
  python bench.py --steps 8 --warmup 3 > $O/bench_c3.json 2> $O/bench_c3.err || echo "bench c3 failed"
  for wl in c1 c2 c4; do
    python bench.py --workload $wl --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_$wl.json 2> $O/bench_$wl.err || echo "bench $wl failed"
  done
  python tools/gemm_ceiling.py > $O/r06_gemm_ceiling.md 2> $O/gemm_ceiling.err || tail -3 $O/gemm_ceiling.err
fi
find $O -name "*.csv" -size +20M -delete
rm -rf $O/prof_c3/*/*.db $O/pmc_fetch $O/pmc_write $O/prof_phase/*/*kernel_trace.csv $O/prof_phase/*/*hip_api_trace.csv
du -sh $O; ls $O
head -30 $O/r06_c3_kernel_stats.md; cat $O/r06_c3_phases.md; cat $O/pmc_summary.txt; cat $O/r06_gemm_ceiling.md
