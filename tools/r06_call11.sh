#!/bin/bash
# round 6, call 11: column bands of the tile order -- correctness (pure renumbering) and C3 in-step A/B (bands rule vs none), 4 interleaved pairs
O=gpurun_out/r06_call11; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_headline_gpu.py -m gpu -q -x -k "renumbering or gemm_at_bench or test_gemm_plain or splitk or split_k or thin_tail" > $O/tests.txt 2>&1; echo "rc $?" >> $O/tests.txt; tail -4 $O/tests.txt
for i in 1 2 3 4; do
  for b in 1 0; do
    SLAM_GEMM_BANDS=$b timeout 400 python bench.py --steps 16 --warmup 4 --no-cpu-baseline > $O/bench_c3_bands${b}_$i.json 2> $O/bench_c3_bands${b}_$i.err || tail -3 $O/bench_c3_bands${b}_$i.err
  done
done
for b in 2 4; do SLAM_GEMM_BANDS=$b timeout 400 python bench.py --steps 16 --warmup 4 --no-cpu-baseline > $O/bench_c3_bands${b}_1.json 2> $O/bench_c3_bands${b}_1.err; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_call11/bench_*.json')):
    d=json.load(open(f)); k=d['kernels']; print(f.split('/')[-1], round(d['ms_per_step'],2), round(d['roofline']['frac'],4), round(k['gemm_nt_w4_kernel<256,256,false,0>']['ms_per_step'],2))
PY
