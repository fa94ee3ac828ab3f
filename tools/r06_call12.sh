#!/bin/bash
# round 6, call 12: block-cyclic dealing of the GEMM tile order to the XCDs -- correctness, then C3 in-step, 3 interleaved rounds
O=gpurun_out/r06_call12; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_headline_gpu.py -m gpu -q -x -k "renumbering or gemm_at_bench or splitk or split_k or thin_tail" > $O/tests.txt 2>&1; echo "rc $?" >> $O/tests.txt; tail -3 $O/tests.txt
for i in 1 2 3; do
  for x in 0 -1 32 256 1; do
    SLAM_GEMM_XBLK=$x timeout 400 python bench.py --steps 16 --warmup 4 --no-cpu-baseline > $O/bench_c3_x${x}_$i.json 2> $O/bench_c3_x${x}_$i.err || tail -3 $O/bench_c3_x${x}_$i.err
  done
done
python - <<'PY'
import json,glob,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r06_call12/bench_*.json')):
    d=json.load(open(f)); x=f.split('_x')[1].split('_')[0]; acc[x].append((d['ms_per_step'], d['roofline']['frac']))
for x,v in acc.items(): print('xblk', x, [round(a,2) for a,_ in v], 'mean', round(sum(a for a,_ in v)/len(v),2), 'frac', round(sum(b for _,b in v)/len(v),4))
PY
