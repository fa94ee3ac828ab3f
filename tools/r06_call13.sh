#!/bin/bash
# round 6, call 13: the opt-in SwiGLU epilogue fusions again, now on top of the raster rule (C3 in-step, 3 interleaved rounds)
O=gpurun_out/r06_call13; mkdir -p $O
for i in 1 2 3; do
  for v in 00 10 01 11; do
    SLAM_FUSED_SWIGLU_FWD=${v:0:1} SLAM_FUSED_SWIGLU_BWD=${v:1:1} timeout 400 python bench.py --steps 16 --warmup 4 --no-cpu-baseline > $O/bench_c3_f${v}_$i.json 2> $O/bench_c3_f${v}_$i.err || tail -3 $O/bench_c3_f${v}_$i.err
  done
done
python - <<'PY'
import json,glob,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r06_call13/bench_*.json')):
    d=json.load(open(f)); x=f.split('_f')[1].split('_')[0]; acc[x].append(d['ms_per_step'])
for x,v in acc.items(): print('fwd,bwd fused =', x, [round(a,2) for a in v], 'mean', round(sum(v)/len(v),2))
PY
