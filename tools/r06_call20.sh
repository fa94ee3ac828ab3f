#!/bin/bash
# round 6, call 20: SwiGLU elementwise passes walking their rows from the END (the producer's freshest lines first; SLAM_EW_REVERSE bit 0 = forward, bit 1 = backward) -- C3 in-step A/B, interleaved
O=gpurun_out/r06_call20; mkdir -p $O
for i in 1 2 3; do
  for e in 0 1 2 3; do
    SLAM_EW_REVERSE=$e timeout 400 python bench.py --steps 16 --warmup 4 --no-cpu-baseline > $O/bench_c3_rev${e}_$i.json 2> $O/bench_c3_rev${e}_$i.err || tail -3 $O/bench_c3_rev${e}_$i.err
  done
done
python - <<'PY'
import json,glob,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r06_call20/bench_c3_rev*.json')):
    d=json.load(open(f)); x=f.split('rev')[1].split('_')[0]
    k=d['kernels']
    acc[x].append((d['ms_per_step'], k['gemm_nt_w4_kernel<256,256,false,0>']['ms_per_step'], d['loss']))
for x,v in sorted(acc.items()): print('rev',x,'ms',[round(a[0],2) for a in v],'w4',[round(a[1],2) for a in v],'loss',v[0][2])
PY
export TMPDIR=/tmp
R=$(pwd)
for e in 0 3; do
  (cd /tmp && SLAM_EW_REVERSE=$e rocprofv3 --kernel-trace --stats -d $R/$O/prof_rev$e -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $R/$O/bench_prof_rev$e.json 2> $R/$O/prof_rev$e.err)
  python tools/rocpd_stats.py $(ls $O/prof_rev$e/*/*.db | head -1) $O/kernel_stats_rev$e.md > /dev/null
  grep -E "swiglu|gemm_nt_w4_kernel<256, 256, false, 0, false, 0>|rmsnorm" $O/kernel_stats_rev$e.md | cut -c1-60,140-220
  rm -rf $O/prof_rev$e
done
