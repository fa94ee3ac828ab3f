#!/bin/bash
# GPU box, round 4, first call: ds_read_b64_tr_b16 probe; PMC passes (MFMA busy / wave cycles / issue stalls / effective clock) on the
# dominant GEMM shapes and the attention kernels; the epilogue A/B of round 3 under GRBM_GUI_ACTIVE; C3 bench line of this box.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04a
mkdir -p $O
export TMPDIR=/tmp
cd $R
python tools/probes/tr_probe.py > $O/tr_probe.json 2> $O/tr_probe.err || echo "tr probe failed"
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"
for what in gemm attn; do
  if [ $what = gemm ]; then CMD="python $R/tools/pmc_gemm.py 5"; else CMD="python $R/tools/attn_xcd_ab.py --only xcd"; fi
  (cd /tmp && rocprofv3 --pmc $P1 --output-format csv -d $O/pmc1_$what -- $CMD > /dev/null 2> $O/pmc1_$what.err)
  (cd /tmp && rocprofv3 --pmc $P2 --output-format csv -d $O/pmc2_$what -- $CMD > /dev/null 2> $O/pmc2_$what.err)
done
# round 3's epilogue A/B: the library before commit 5ed10e8 (17.2 k probe cycles per tile) against HEAD (10.5 k)
if [ -f $R/tools/probes/libslamhip_pre_epi.so ]; then
  (cd /tmp && SLAM_HIP_LIB=$R/tools/probes/libslamhip_pre_epi.so rocprofv3 --pmc $P1 --output-format csv -d $O/pmc1_gemm_pre_epi -- python $R/tools/pmc_gemm.py 5 > /dev/null 2> $O/pmc1_gemm_pre_epi.err)
  (cd /tmp && rocprofv3 --pmc $P1 --output-format csv -d $O/pmc1_gemm_head2 -- python $R/tools/pmc_gemm.py 5 > /dev/null 2> $O/pmc1_gemm_head2.err)
fi
python tools/pmc_table.py $O/pmc_gemm.md "GEMM, HEAD (auto rule)" $O/pmc1_gemm $O/pmc2_gemm > /dev/null 2> $O/table.err
python tools/pmc_table.py $O/pmc_attn.md "attention kernels, HEAD" $O/pmc1_attn $O/pmc2_attn > /dev/null 2>> $O/table.err
[ -d $O/pmc1_gemm_pre_epi ] && python tools/pmc_table.py $O/pmc_gemm_epi_ab.md "GEMM epilogue A/B: before 5ed10e8 | HEAD (second pass)" $O/pmc1_gemm_pre_epi > /dev/null 2>> $O/table.err
[ -d $O/pmc1_gemm_head2 ] && python tools/pmc_table.py $O/pmc_gemm_epi_ab_head.md "GEMM epilogue A/B: HEAD (second pass)" $O/pmc1_gemm_head2 > /dev/null 2>> $O/table.err
python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err || echo "bench failed"
find $O -name "*.csv" -size +20M -delete
du -sh $O; ls $O
cat $O/tr_probe.json | head -60
