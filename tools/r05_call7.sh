#!/bin/bash
# round 5, GPU call 7: attention built without SLP packing, P kept where the second product reads it, pre-scaled-Q form of the frozen
# Whisper forward: parity of every attention test, kernel A/B against the baseline build (slam_llm_amd/libslamhip_base.so), step A/B
O=gpurun_out/r05g
mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_headline_gpu.py -q -x -k "attn or attention" -p no:cacheprovider > $O/attn_tests.log 2>&1
echo "rc $?" >> $O/attn_tests.log
tail -4 $O/attn_tests.log
for i in 1 2; do
  SLAM_HIP_LIB=$PWD/slam_llm_amd/libslamhip_base.so timeout 300 python tools/attn_lib_ab.py >> $O/attn_lib_ab.jsonl 2>> $O/attn_lib_ab.err
  timeout 300 python tools/attn_lib_ab.py >> $O/attn_lib_ab.jsonl 2>> $O/attn_lib_ab.err
done
cat $O/attn_lib_ab.jsonl
for i in 1 2; do
  SLAM_HIP_LIB=$PWD/slam_llm_amd/libslamhip_base.so timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_c3_base_$i.json 2>> $O/bench_err.txt
  timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_c3_new_$i.json 2>> $O/bench_err.txt
done
python - <<'PY'
import json
for k in ("base_1","new_1","base_2","new_2"):
    d=json.load(open(f"gpurun_out/r05g/bench_c3_{k}.json"))
    att={n[:14]:round(v["ms_per_step"],2) for n,v in d["kernels"].items() if "attn" in n}
    print(k, round(d["ms_per_step"],2), round(d["loss"],5), att)
PY
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_geometry_gpu.py -q -x -p no:cacheprovider > $O/model_tests.log 2>&1
echo "rc $?" >> $O/model_tests.log
tail -4 $O/model_tests.log
