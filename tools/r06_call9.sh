#!/bin/bash
# round 6, call 9: in-step sweep of the raster rule's values on C3 (one box, one pass each, then the two best again)
O=gpurun_out/r06_call9; mkdir -p $O
for rule in 12,4,4,8 12,4,4,4 12,4,4,6 12,4,4,12 12,4,2,8 12,4,6,8 12,2,4,8 16,4,4,8 8,4,4,8 12,4,4,8; do
  SLAM_GEMM_GROUP_M_RULE=$rule timeout 400 python bench.py --steps 16 --warmup 4 --no-cpu-baseline > $O/b_$rule.json 2> $O/b_$rule.err || tail -3 $O/b_$rule.err
  python - <<PY
import json
d=json.load(open('$O/b_$rule.json')); print('$rule', round(d['ms_per_step'],2), round(d['roofline']['frac'],4))
PY
done
