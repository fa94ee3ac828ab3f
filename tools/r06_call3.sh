#!/bin/bash
# round 6, call 3: graph-capture tests, the full-depth emulation test, graph on / off A/B on C3 / C1 / C4 / C2
O=gpurun_out/r06_call3; mkdir -p $O
export SLAM_TEST_REPORT=$O/report
timeout 900 python -m pytest tests/test_graph_gpu.py -m gpu -q > $O/graph_tests.txt 2>&1; echo "graph tests rc $?" >> $O/graph_tests.txt
tail -8 $O/graph_tests.txt
timeout 2400 python -m pytest tests/test_headline_gpu.py -m gpu -x -q -k "full_depth" -s > $O/emu_test.txt 2>&1; echo "emulation test rc $?" >> $O/emu_test.txt
grep -E "^family|^q_proj|^v_proj|^projector|^full depth|passed|failed|rc |Error|assert" $O/emu_test.txt | tail -30
for wl in c3 c1 c4 c2; do
  for g in off on; do
    timeout 500 python bench.py --workload $wl --graph $g --steps 12 --warmup 3 --no-cpu-baseline > $O/bench_${wl}_graph_$g.json 2> $O/bench_${wl}_graph_$g.err || tail -5 $O/bench_${wl}_graph_$g.err
  done
done
python - <<'PY'
import json
for wl in ("c3","c1","c4","c2"):
    for g in ("off","on"):
        try:
            d=json.load(open(f'gpurun_out/r06_call3/bench_{wl}_graph_{g}.json'))
            print(wl, g, round(d['ms_per_step'],2), round(d['value'],1), round(d['roofline']['frac'],4), d['config'].get('step_issue'), round(d['loss'],5))
        except Exception as e: print(wl, g, 'failed', e)
PY
