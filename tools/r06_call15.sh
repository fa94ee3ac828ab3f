#!/bin/bash
# round 6, call 15: frozen encoder of the next batch on a low-priority side stream -- bit-identity test, then C3 in-step A/B (3 interleaved pairs)
O=gpurun_out/r06_call15; mkdir -p $O
timeout 600 python -m pytest tests/test_graph_gpu.py -m gpu -q -x -k "encoder_ahead" > $O/tests.txt 2>&1; echo "rc $?" >> $O/tests.txt; tail -3 $O/tests.txt
for i in 1 2 3; do
  timeout 400 python bench.py --steps 16 --warmup 4 --no-cpu-baseline > $O/bench_c3_plain_$i.json 2> $O/bench_c3_plain_$i.err || tail -3 $O/bench_c3_plain_$i.err
  timeout 400 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --encoder-ahead > $O/bench_c3_ahead_$i.json 2> $O/bench_c3_ahead_$i.err || tail -3 $O/bench_c3_ahead_$i.err
done
python - <<'PY'
import json,glob,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r06_call15/bench_*.json')):
    d=json.load(open(f)); x=f.split('bench_c3_')[1].split('_')[0]; acc[x].append((d['ms_per_step'], d['roofline']['frac'], d['loss']))
for x,v in acc.items(): print(x, [round(a,2) for a,_,_ in v], 'mean', round(sum(a for a,_,_ in v)/len(v),2), 'frac', [round(b,4) for _,b,_ in v], 'loss', v[0][2])
PY
