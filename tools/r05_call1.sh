#!/bin/bash
# round 5, GPU call 1: the whole GPU suite (no -x, durations), the WavLM train-mode seed table, a C3 bench line on the same box
O=gpurun_out/r05a
mkdir -p $O
python -m pytest tests -m gpu -q --durations=25 -p no:cacheprovider > $O/gpu_suite.log 2>&1
echo "suite rc $?" >> $O/gpu_suite.log
tail -5 $O/gpu_suite.log
python tools/wavlm_trainmode_seeds.py --seeds 5 --out $O/wavlm_trainmode_seeds.md > $O/seeds.log 2>&1 || tail -20 $O/seeds.log
python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err || tail -5 $O/bench_c3.err
python -c "import json;d=json.load(open('$O/bench_c3.json'));print(d['ms_per_step'],d['value'],d['roofline']['frac'])"
