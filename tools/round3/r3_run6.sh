#!/bin/bash
OUT=gpurun_out/${1:-r3k}
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_engine.py -q -m gpu --timeout 600 -p no:cacheprovider > $OUT/tests.log 2>&1; grep -E "passed|failed" $OUT/tests.log; grep -E "^FAILED|^ERROR|^E " $OUT/tests.log | head -10
