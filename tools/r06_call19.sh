#!/bin/bash
# round 6, call 19: streaming (nt) stores / hoisted residual loads of the bf16 GEMM epilogue (SLAM_GEMM_EPI bits 4 / 2) -- C3 in-step A/B, interleaved
O=gpurun_out/r06_call19; mkdir -p $O
for i in 1 2 3; do
  for e in 0 4 6; do
    SLAM_GEMM_EPI=$e timeout 400 python bench.py --steps 16 --warmup 4 --no-cpu-baseline > $O/bench_c3_epi${e}_$i.json 2> $O/bench_c3_epi${e}_$i.err || tail -3 $O/bench_c3_epi${e}_$i.err
  done
done
python - <<'PY'
import json,glob,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r06_call19/bench_c3_epi*.json')):
    d=json.load(open(f)); x=f.split('epi')[1].split('_')[0]
    k=d['kernels']
    acc[x].append((d['ms_per_step'], k['gemm_nt_w4_kernel<256,256,false,0>']['ms_per_step'], k['gemm_nt_persist2_kernel<256,256,2,4>']['ms_per_step'], k['gemm_nt_pipe_kernel<256,256,2,4,1>']['ms_per_step'], d['loss']))
for x,v in sorted(acc.items()): print('epi',x,'ms',[round(a[0],2) for a in v],'w4',[round(a[1],2) for a in v],'persist2',[round(a[2],2) for a in v],'pipe',[round(a[3],2) for a in v],'loss',v[0][4])
PY
