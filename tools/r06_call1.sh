#!/bin/bash
# round 6, call 1: new tests (graph capture, label rows, QS tail, logits at true widths) + a C3 bench line at HEAD
O=gpurun_out/r06_call1; mkdir -p $O
export SLAM_TEST_REPORT=$O/report
timeout 900 python -m pytest tests/test_graph_gpu.py tests/test_ops_gpu.py -m gpu -x -q -k "graph or label_rows or adamw_step_dev or static_label or prescaled or test_gemm_plain" > $O/new_tests.txt 2>&1; echo "new tests rc $?" >> $O/new_tests.txt
tail -15 $O/new_tests.txt
timeout 1500 python -m pytest tests/test_headline_gpu.py tests/test_geometry_gpu.py -m gpu -x -q -k "headline_geometry or c4_true or c1_true or c4_bench or c5_style or full_depth" -s > $O/logits_tests.txt 2>&1; echo "logits tests rc $?" >> $O/logits_tests.txt
grep -E "^logits|passed|failed|rc |Error|DRIFT|worst gradient" $O/logits_tests.txt | tail -30
timeout 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06_call1/bench_c3.json'))
print(d['ms_per_step'], d['value'], d['roofline']['frac'])
PY
