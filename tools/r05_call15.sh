#!/bin/bash
# round 5, GPU call 15: kernel tables of the secondary workloads at HEAD (C4: HuBERT-large -> Vicuna-7B Q-Former; C2: Whisper-base ->
# Llama-3-8B; C1) incl. the inter-kernel idle share -- what the rest of their step is made of
O=gpurun_out/r05m
mkdir -p $O
R=$PWD
for wl in c4 c2 c1; do
  (cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$wl -- python $R/bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline > $R/$O/prof_$wl.json 2> $R/$O/prof_$wl.err)
  python tools/rocpd_stats.py $(ls $O/prof_$wl/*/*.db | head -1) $O/r05_${wl}_kernel_stats.md > /dev/null 2>&1
  rm -rf $O/prof_$wl
  python -c "import json;d=json.load(open('$O/prof_$wl.json'));print('$wl',round(d['ms_per_step'],2),'ms under the profiler')"
  head -16 $O/r05_${wl}_kernel_stats.md | cut -c1-170
  tail -3 $O/r05_${wl}_kernel_stats.md
done
