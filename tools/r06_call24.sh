#!/bin/bash
# round 6, call 24: LayerNorm with two rows per wave for narrow rows (SLAM_LN_NARROW=2; 0 = one row per wave) -- C3 in-step A/B (interleaved) + the kernel's own average under rocprofv3
O=gpurun_out/r06_call24; mkdir -p $O
for i in 1 2 3; do
  for e in 0 2; do
    SLAM_LN_NARROW=$e timeout 400 python bench.py --steps 16 --warmup 4 --no-cpu-baseline > $O/bench_c3_ln${e}_$i.json 2> $O/bench_c3_ln${e}_$i.err || tail -3 $O/bench_c3_ln${e}_$i.err
  done
done
python - <<'PY'
import json,glob,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r06_call24/bench_c3_ln*.json')):
    d=json.load(open(f)); x=f.split('_ln')[1].split('_')[0]
    acc[x].append(d['ms_per_step'])
for x,v in sorted(acc.items()): print('ln_narrow',x,'ms',[round(a,2) for a in v],'mean',round(sum(v)/len(v),2))
PY
export TMPDIR=/tmp
R=$(pwd)
for e in 0 2; do
  (cd /tmp && SLAM_LN_NARROW=$e rocprofv3 --kernel-trace --stats -d $R/$O/prof_ln$e -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $R/$O/bench_prof_ln$e.json 2> $R/$O/prof_ln$e.err)
  python tools/rocpd_stats.py $(ls $O/prof_ln$e/*/*.db | head -1) $O/kernel_stats_ln$e.md > /dev/null
  grep -E "layernorm" $O/kernel_stats_ln$e.md | cut -c1-70,150-240
  rm -rf $O/prof_ln$e
done
