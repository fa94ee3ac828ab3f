#!/bin/bash
# round 6, call 14: SwiGLU forward in the gate|up epilogue on top of the raster rule -- confirmation on another box, 4 interleaved pairs; C2 / C4 too
O=gpurun_out/r06_call14; mkdir -p $O
for i in 1 2 3 4; do
  for v in 0 1; do
    SLAM_FUSED_SWIGLU_FWD=$v timeout 400 python bench.py --steps 16 --warmup 4 --no-cpu-baseline > $O/bench_c3_fwd${v}_$i.json 2> $O/bench_c3_fwd${v}_$i.err || tail -3 $O/bench_c3_fwd${v}_$i.err
  done
done
for wl in c2 c4 c1; do for v in 0 1; do
  SLAM_FUSED_SWIGLU_FWD=$v timeout 300 python bench.py --workload $wl --steps 16 --warmup 4 --no-cpu-baseline > $O/bench_${wl}_fwd${v}.json 2> $O/bench_${wl}_fwd${v}.err
done; done
python - <<'PY'
import json,glob,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r06_call14/bench_*.json')):
    d=json.load(open(f)); x=f.split('bench_')[1].rsplit('_',1)[0] if 'c3' in f else f.split('bench_')[1][:-5]; acc[x].append(d['ms_per_step'])
for x,v in sorted(acc.items()): print(x, [round(a,2) for a in v], 'mean', round(sum(v)/len(v),2))
PY
