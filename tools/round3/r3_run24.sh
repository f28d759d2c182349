#!/bin/bash
OUT=gpurun_out/${1:-r3G}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_quant.py tests/test_gpu_tp_shards.py tests/test_gpu_parity_qwen3.py -m gpu -x -q > $OUT/tests.log 2>&1; grep -n "passed\|failed" $OUT/tests.log; grep -n "^FAILED\|^ERROR\|Error" $OUT/tests.log | head
