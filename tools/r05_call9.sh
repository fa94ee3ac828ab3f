#!/bin/bash
# round 5, GPU call 9: the 32-keys-per-wave dK / dV kernel: bit identity + every attention test, kernel A/B, step A/B
O=gpurun_out/r05i
mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_headline_gpu.py -q -x -k "attn or attention" -p no:cacheprovider > $O/attn_tests.log 2>&1
echo "rc $?" >> $O/attn_tests.log
tail -4 $O/attn_tests.log
timeout 300 python tools/attn_dkdv_ab.py > $O/attn_dkdv_ab.json 2> $O/attn_dkdv_ab.err
cat $O/attn_dkdv_ab.err | tail -6
for i in 1 2; do
  SLAM_ATTN_DKDV32=0 timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_c3_k16_$i.json 2>> $O/bench_err.txt
  timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_c3_k32_$i.json 2>> $O/bench_err.txt
done
python - <<'PY'
import json
for k in ("k16_1","k32_1","k16_2","k32_2"):
    d=json.load(open(f"gpurun_out/r05i/bench_c3_{k}.json"))
    att={n[:14]:round(v["ms_per_step"],2) for n,v in d["kernels"].items() if "attn" in n}
    print(k, round(d["ms_per_step"],2), round(d["loss"],5), att)
PY
