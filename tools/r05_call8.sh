#!/bin/bash
# round 5, GPU call 8: what the label-rows / every-row gradient comparison measures on the baseline build and on the current one
O=gpurun_out/r05h
mkdir -p $O
SLAM_HIP_LIB=$PWD/slam_llm_amd/libslamhip_base.so timeout 600 python -m pytest tests/test_model_gpu.py -q -s -k "label_rows_equals" -p no:cacheprovider 2>&1 | grep "label-rows\|passed\|failed" > $O/label_rows_base.txt
timeout 600 python -m pytest tests/test_model_gpu.py -q -s -k "label_rows_equals" -p no:cacheprovider 2>&1 | grep "label-rows\|passed\|failed" > $O/label_rows_new.txt
echo BASE; cat $O/label_rows_base.txt; echo NEW; cat $O/label_rows_new.txt
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_geometry_gpu.py tests/test_boundary_gpu.py -q -p no:cacheprovider --deselect "tests/test_model_gpu.py::test_lm_head_over_label_rows_equals_every_row" > $O/model_tests.log 2>&1
echo "rc $?" >> $O/model_tests.log
tail -6 $O/model_tests.log
