#!/bin/bash
O=gpurun_out/r05e
mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py -q -k "ragged or unfrozen" -p no:cacheprovider > $O/tests.log 2>&1
tail -30 $O/tests.log
