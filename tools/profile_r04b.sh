#!/bin/bash
# GPU box, round 4, HEAD: PMC passes (MFMA busy / wave cycles / effective clock / instruction mix) on the attention kernels that are the
# default now (transposed-read forms) and on the log-mel kernel.  Everything lands under gpurun_out/r04b/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04b
mkdir -p $O
export TMPDIR=/tmp
cd $R
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"
for what in attn logmel; do
  if [ $what = attn ]; then CMD="python $R/tools/attn_xcd_ab.py --only xcd"; else CMD="python $R/tools/logmel_bench.py"; fi
  (cd /tmp && rocprofv3 --pmc $P1 --output-format csv -d $O/pmc1_$what -- $CMD > /dev/null 2> $O/pmc1_$what.err)
  (cd /tmp && rocprofv3 --pmc $P2 --output-format csv -d $O/pmc2_$what -- $CMD > /dev/null 2> $O/pmc2_$what.err)
  python tools/pmc_table.py $O/pmc_$what.md "$what kernels, HEAD (round 4, transposed-read attention / folded log-mel)" $O/pmc1_$what $O/pmc2_$what > $O/table_$what.json 2>> $O/table.err
done
find $O -name "*.csv" -size +20M -delete
rm -rf $O/pmc1_*/*/*.db $O/pmc2_*/*/*.db
du -sh $O; cat $O/pmc_attn.md | cut -c1-400 | head -12; cat $O/pmc_logmel.md | cut -c1-400 | head -12
