#!/bin/bash
# round 5, GPU call 6 (HEAD after the ragged un-frozen Whisper commit): the driver's GPU suite (-x), smoke, the C3 kernel table under
# rocprofv3 and an un-profiled C3 bench line on the same box
O=gpurun_out/r05f
mkdir -p $O
R=$PWD
SLAM_TEST_MARGINS=$O/margins.tsv SLAM_TEST_REPORT=$O/c3_full_depth.txt timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=15 -p no:cacheprovider > $O/gpu_suite.log 2>&1
echo "rc $?" >> $O/gpu_suite.log
tail -22 $O/gpu_suite.log
python tools/margins_report.py $O/margins.tsv > $O/margins.md 2>/dev/null; head -12 $O/margins.md
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_c3 -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $R/$O/prof_c3.json 2> $R/$O/prof_c3.err)
python tools/rocpd_stats.py $(ls $O/prof_c3/*/*.db | head -1) $O/r05_c3_kernel_stats.md > /dev/null 2>&1
rm -rf $O/prof_c3
timeout 600 python bench.py --steps 10 --warmup 3 > $O/bench_c3.json 2> $O/bench_c3.err || tail -5 $O/bench_c3.err
python -c "import json;d=json.load(open('$O/bench_c3.json'));print(d['ms_per_step'],d['value'],d['roofline']['frac'],d['cpu_baseline']['value'])"
head -24 $O/r05_c3_kernel_stats.md | cut -c1-200
