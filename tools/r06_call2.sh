#!/bin/bash
# round 6, call 2: graph-capture tests, the full-depth emulation test, graph A/B on C1/C3/C4/C2
O=gpurun_out/r06_call2; mkdir -p $O
export SLAM_TEST_REPORT=$O/report
timeout 900 python -m pytest tests/test_graph_gpu.py -m gpu -q > $O/graph_tests.txt 2>&1; echo "graph tests rc $?" >> $O/graph_tests.txt
tail -25 $O/graph_tests.txt
timeout 2400 python -m pytest tests/test_headline_gpu.py -m gpu -x -q -k "full_depth" -s > $O/emu_test.txt 2>&1; echo "emulation test rc $?" >> $O/emu_test.txt
grep -E "^family|^q_proj|^v_proj|^projector|^full depth|passed|failed|rc |Error|assert" $O/emu_test.txt | tail -30
