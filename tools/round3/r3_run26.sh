#!/bin/bash
OUT=gpurun_out/${1:-r3I}
mkdir -p $OUT
timeout 300 python tools/prefill_sweep.py 0 1024,2048 2>&1 | tail -3
timeout 300 python tools/bench_vit_batch.py qwen3-vl-2b 1 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_parity_headline.py tests/test_golden_qwen3.py tests/test_qwen3_vl.py tests/test_qwen3_5.py tests/test_gpu_parity_qwen3.py -m gpu -x -q > $OUT/tests.log 2>&1; grep -n "passed\|failed" $OUT/tests.log; grep -n "^FAILED\|^ERROR\|Error" $OUT/tests.log | head
