#!/bin/bash
# round 6, call 16: start stagger of the persistent GEMM (needs profiles/r06_gemm_stagger.patch applied; tools/gemm_stagger.py was tools/gemm_enc_raster.py with the group-height knob replaced by slam_gemm_set_config(1000 + percent) / (1300 + mode) and a bit-identity check) -- isolated sweep, then C3 in-step A/B
O=gpurun_out/r06_call16; mkdir -p $O
timeout 600 python tools/gemm_stagger.py > $O/gemm_stagger.jsonl 2> $O/gemm_stagger.err; echo "rc $?"; cat $O/gemm_stagger.jsonl | cut -c1-900
for i in 1 2; do
  for s in 0 50 100; do
    SLAM_GEMM_STAGGER=$s timeout 400 python bench.py --steps 16 --warmup 4 --no-cpu-baseline > $O/bench_c3_stag${s}_$i.json 2> $O/bench_c3_stag${s}_$i.err || tail -3 $O/bench_c3_stag${s}_$i.err
  done
done
python - <<'PY'
import json,glob,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r06_call16/bench_c3_stag*.json')):
    d=json.load(open(f)); x=f.split('stag')[1].split('_')[0]
    k=d['kernels']; acc[x].append((d['ms_per_step'], k['gemm_nt_persist2_kernel<256,256,2,4>']['ms_per_step'], d['loss']))
for x,v in acc.items(): print('stagger',x, 'ms', [round(a,2) for a,_,_ in v], 'persist2 ms', [round(b,2) for _,b,_ in v], 'loss', v[0][2])
PY
