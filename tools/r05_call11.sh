#!/bin/bash
# round 5, GPU call 11: the softmax scale folded into the frozen Whisper query projection (negative scale at the C ABI): op tests, the
# model tests that run the frozen Whisper encoder, smoke, one bench line
O=gpurun_out/r05k
mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "attn or attention" -p no:cacheprovider > $O/attn_tests.log 2>&1
echo "rc $?" >> $O/attn_tests.log
tail -3 $O/attn_tests.log
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_geometry_gpu.py tests/test_boundary_gpu.py tests/test_headline_gpu.py -q -x -p no:cacheprovider -k "not full_depth" > $O/model_tests.log 2>&1
echo "rc $?" >> $O/model_tests.log
tail -4 $O/model_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err
python -c "import json;d=json.load(open('$O/bench_c3.json'));print(d['ms_per_step'],d['value'],d['roofline']['frac'],d['loss'])"
