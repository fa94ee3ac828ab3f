#!/bin/bash
# round 6, call 4: graph + side-stream tests, emulation test, side-stream A/B on C3 (interleaved), C1 with one eager step in the timed region
O=gpurun_out/r06_call4; mkdir -p $O
export SLAM_TEST_REPORT=$O/report
timeout 900 python -m pytest tests/test_graph_gpu.py -m gpu -q > $O/graph_tests.txt 2>&1; echo "graph tests rc $?" >> $O/graph_tests.txt
tail -8 $O/graph_tests.txt
timeout 2400 python -m pytest tests/test_headline_gpu.py -m gpu -x -q -k "full_depth" -s > $O/emu_test.txt 2>&1; echo "emulation test rc $?" >> $O/emu_test.txt
grep -E "^family|^q_proj|^v_proj|^projector|^full depth|passed|failed|rc |Error|assert" $O/emu_test.txt | tail -30
for i in 1 2; do
  for s in 0 1; do
    SLAM_LORA_SIDE_STREAM=$s timeout 500 python bench.py --steps 12 --warmup 3 --no-cpu-baseline > $O/bench_c3_side${s}_$i.json 2> $O/bench_c3_side${s}_$i.err || tail -5 $O/bench_c3_side${s}_$i.err
  done
done
SLAM_BENCH_TIMING_EVERY=40 timeout 300 python bench.py --workload c1 --graph on --steps 40 --warmup 3 --no-cpu-baseline > $O/bench_c1_graph_on_40.json 2> $O/c1a.err
SLAM_BENCH_TIMING_EVERY=40 timeout 300 python bench.py --workload c1 --graph off --steps 40 --warmup 3 --no-cpu-baseline > $O/bench_c1_graph_off_40.json 2> $O/c1b.err
SLAM_LORA_SIDE_STREAM=1 SLAM_BENCH_TIMING_EVERY=40 timeout 300 python bench.py --workload c1 --graph on --steps 40 --warmup 3 --no-cpu-baseline > $O/bench_c1_graph_on_side_40.json 2> $O/c1c.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_call4/bench_*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], round(d['ms_per_step'],2), round(d['value'],1), round(d['roofline']['frac'],4), d['config'].get('step_issue','')[:40], round(d['loss'],5))
    except Exception as e: print(f, 'failed', e)
PY
