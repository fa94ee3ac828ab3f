#!/bin/bash
# round 6, call 22: (a) HIP_FORCE_DEV_KERNARG set before / after `import torch` (launch-rate probe); (b) more runtime knobs on top of it, C4 + C3 in-step, interleaved
O=gpurun_out/r06_call22; mkdir -p $O
for m in unset before after unset before after; do python tools/kernarg_probe.py $m; done 2>&1 | grep -v amdgpu.ids | tee $O/kernarg_probe.txt
for i in 1 2 3; do
  for wl in c4 c3; do
    for v in base intr scratch q1 q8; do
      case $v in
        base) E="";;
        intr) E="HSA_ENABLE_INTERRUPT=0";;
        scratch) E="HSA_NO_SCRATCH_RECLAIM=1";;
        q1) E="GPU_MAX_HW_QUEUES=1";;
        q8) E="GPU_MAX_HW_QUEUES=8";;
      esac
      env $E timeout 400 python bench.py --workload $wl --steps 16 --warmup 4 --no-cpu-baseline > $O/bench_${wl}_${v}_$i.json 2> $O/bench_${wl}_${v}_$i.err || tail -3 $O/bench_${wl}_${v}_$i.err
    done
  done
done
python - <<'PY'
import json,glob,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r06_call22/bench_*.json')):
    d=json.load(open(f)); n=f.split('bench_')[1].split('_'); acc[(n[0],n[1])].append(d['ms_per_step'])
for x,v in sorted(acc.items()): print(x,[round(a,2) for a in v],'mean',round(sum(v)/len(v),3))
PY
