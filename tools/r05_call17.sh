#!/bin/bash
# round 5, GPU call 17: PMC passes (separate --pmc runs, no trace domains) over the attention kernels at HEAD
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05q
mkdir -p $O
export TMPDIR=/tmp
cd $R
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"
CMD="python $R/tools/pmc_attn_r05.py 4"
(cd /tmp && timeout 300 rocprofv3 --pmc $P1 --output-format csv -d $O/pmc1 -- $CMD > /dev/null 2> $O/pmc1.err)
(cd /tmp && timeout 300 rocprofv3 --pmc $P2 --output-format csv -d $O/pmc2 -- $CMD > /dev/null 2> $O/pmc2.err)
python tools/pmc_table.py $O/pmc_attn.md "attention kernels at HEAD (round 5): accumulators-from--m Whisper forward vs general softmax, 16-key vs 32-key dK / dV" $O/pmc1 $O/pmc2 > $O/table.json 2> $O/table.err
find $O -name "*.csv" -size +20M -delete
rm -rf $O/pmc1/*/*.db $O/pmc2/*/*.db
cat $O/pmc_attn.md | cut -c1-600 | head -20
