#!/bin/bash
# round 5, GPU call 2: the two-workgroups-per-CU GEMM (parity, then timing), C3 at full depth with the gradient table
O=gpurun_out/r05b
mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "gemm" -p no:cacheprovider > $O/gemm_tests.log 2>&1
echo "gemm tests rc $?" >> $O/gemm_tests.log
tail -4 $O/gemm_tests.log
timeout 400 python tools/gemm_p3_bench.py > $O/p3_bench.jsonl 2> $O/p3_bench.err || tail -5 $O/p3_bench.err
cat $O/p3_bench.jsonl
SLAM_TEST_REPORT=$O/c3_full_depth.txt timeout 900 python -m pytest tests/test_headline_gpu.py -q -k full_depth -p no:cacheprovider > $O/full_depth.log 2>&1
tail -3 $O/full_depth.log
cat $O/c3_full_depth.txt
