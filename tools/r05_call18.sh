#!/bin/bash
# round 5, GPU call 18: exponentials of the QS forward placed between the MFMAs of the first product: parity, kernel timing (knob 61 vs 60)
O=gpurun_out/r05r
mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "attn or attention" -p no:cacheprovider > $O/attn_tests.log 2>&1
echo "rc $?" >> $O/attn_tests.log
tail -3 $O/attn_tests.log
timeout 300 python tools/attn_lib_ab.py > $O/attn_lib_ab.jsonl 2> $O/attn_lib_ab.err; cat $O/attn_lib_ab.jsonl
timeout 300 python tools/attn_lib_ab.py >> $O/attn_lib_ab.jsonl 2>> $O/attn_lib_ab.err; tail -1 $O/attn_lib_ab.jsonl
timeout 600 python -m pytest tests/test_model_gpu.py tests/test_headline_gpu.py -q -x -p no:cacheprovider -k "reference or ragged_encoder or headline_geometry or smoke or edge_shapes" > $O/model_tests.log 2>&1
echo "rc $?" >> $O/model_tests.log
tail -3 $O/model_tests.log
