#!/bin/bash
# round 5, GPU call 4: heaviest-block-first attention order: parity, kernel A/B, step A/B
O=gpurun_out/r05d
mkdir -p $O
timeout 600 python -m pytest tests/test_headline_gpu.py tests/test_ops_gpu.py -q -k "attn or attention" -p no:cacheprovider > $O/attn_tests.log 2>&1
tail -3 $O/attn_tests.log
timeout 300 python tools/attn_heavy_ab.py > $O/attn_heavy_ab.json 2> $O/attn_heavy_ab.err
cat $O/attn_heavy_ab.err
for i in 1 2; do
  SLAM_ATTN_HEAVY=0 timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_c3_idorder_$i.json 2> $O/bench_err.txt
  timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_c3_heavy_$i.json 2>> $O/bench_err.txt
done
python - <<'PY'
import json
for k in ("idorder_1","heavy_1","idorder_2","heavy_2"):
    d=json.load(open(f"gpurun_out/r05d/bench_c3_{k}.json"))
    ks=d["kernels"]
    att={n[:40]:round(v["ms_per_step"],2) for n,v in ks.items() if "attn" in n}
    print(k, round(d["ms_per_step"],2), att)
PY
