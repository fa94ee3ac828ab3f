#!/bin/bash
# round 6, call 10: raster evidence -- isolated sweeps (encoder + LLM products) saved, then 4 interleaved in-step pairs rule vs one-size raster
O=gpurun_out/r06_call10; mkdir -p $O
timeout 600 python tools/gemm_enc_raster.py > $O/raster_enc.jsonl 2> $O/raster_enc.err
timeout 800 python tools/gemm_enc_raster.py llm > $O/raster_llm.jsonl 2> $O/raster_llm.err
for i in 1 2 3 4; do
  for g in 8 0; do
    SLAM_GEMM_GROUP_M=$g timeout 400 python bench.py --steps 16 --warmup 4 --no-cpu-baseline > $O/bench_c3_gm${g}_$i.json 2> $O/bench_c3_gm${g}_$i.err || tail -3 $O/bench_c3_gm${g}_$i.err
  done
done
for wl in c2 c4; do
  for g in 8 0; do
    SLAM_GEMM_GROUP_M=$g timeout 300 python bench.py --workload $wl --steps 16 --warmup 4 --no-cpu-baseline > $O/bench_${wl}_gm${g}.json 2> $O/bench_${wl}_gm${g}.err
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_call10/bench_*.json')):
    d=json.load(open(f)); print(f.split('/')[-1], round(d['ms_per_step'],2), round(d['roofline']['frac'],4))
PY
