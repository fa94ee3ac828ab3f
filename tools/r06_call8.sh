#!/bin/bash
# round 6, call 8: per-shape raster rule (default) vs the one-size raster of rounds 1-5 (SLAM_GEMM_GROUP_M=8), C3 in-step, interleaved
O=gpurun_out/r06_call8; mkdir -p $O
for i in 1 2 3; do
  for g in 8 0; do
    SLAM_GEMM_GROUP_M=$g timeout 400 python bench.py --steps 16 --warmup 4 --no-cpu-baseline > $O/bench_c3_gm${g}_$i.json 2> $O/bench_c3_gm${g}_$i.err || tail -3 $O/bench_c3_gm${g}_$i.err
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_call8/bench_*.json')):
    d=json.load(open(f)); k=d['kernels']
    print(f.split('/')[-1], round(d['ms_per_step'],2), round(d['roofline']['frac'],4), {n[:28]: round(v['ms_per_step'],2) for n,v in k.items() if n.startswith('gemm_nt_w4_kernel<256,256,false,0>') or 'persist2' in n or 'pipe' in n})
PY
