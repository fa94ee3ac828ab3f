#!/bin/bash
# round 6: the whole GPU suite exactly as the driver runs it (-x -q -m gpu), raw log + cosine margins + drift report kept; then smoke()
O=gpurun_out/r06v
mkdir -p $O
SLAM_TEST_MARGINS=$O/margins.tsv SLAM_TEST_REPORT=$O/report python -m pytest tests/ -x -q -m gpu --durations=15 -p no:cacheprovider > $O/gpu_suite.log 2>&1
echo "rc $?" >> $O/gpu_suite.log
tail -45 $O/gpu_suite.log
python tools/margins_report.py $O/margins.tsv > $O/margins.md
head -16 $O/margins.md
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
