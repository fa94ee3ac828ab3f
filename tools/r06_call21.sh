#!/bin/bash
# round 6, call 21: HIP runtime knobs in-step (HIP_FORCE_DEV_KERNARG: kernel arguments written straight to device memory) -- C3 / C1 / C4 / C2, interleaved
O=gpurun_out/r06_call21; mkdir -p $O
for i in 1 2 3; do
  for wl in c3 c1 c4 c2; do
    for k in 0 1; do
      HIP_FORCE_DEV_KERNARG=$k timeout 400 python bench.py --workload $wl --steps 16 --warmup 4 --no-cpu-baseline > $O/bench_${wl}_ka${k}_$i.json 2> $O/bench_${wl}_ka${k}_$i.err || tail -3 $O/bench_${wl}_ka${k}_$i.err
    done
  done
done
python - <<'PY'
import json,glob,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r06_call21/bench_*_ka*.json')):
    d=json.load(open(f)); n=f.split('bench_')[1]; wl=n.split('_')[0]; k=n.split('_ka')[1][0]
    acc[(wl,k)].append(d['ms_per_step'])
for x,v in sorted(acc.items()): print(x,[round(a,2) for a in v],'mean',round(sum(v)/len(v),3))
PY
