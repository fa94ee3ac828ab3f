#!/bin/bash
# round 6, call 7: the three tests added after the profile set (Join with the real model, true-width trajectory, true-width graph capture)
O=gpurun_out/r06_call7; mkdir -p $O
timeout 1500 python -m pytest tests/test_dist_gpu.py tests/test_graph_gpu.py tests/test_headline_gpu.py -m gpu -q -s -k "two_ranks or true_widths or trajectory" > $O/tests.txt 2>&1; echo "rc $?" >> $O/tests.txt
grep -E "passed|failed|rc |Error|assert|trajectory" $O/tests.txt | tail -20
