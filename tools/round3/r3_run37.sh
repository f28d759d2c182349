#!/bin/bash
OUT=gpurun_out/${1:-r3U}
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_engine.py tests/test_gpu_parity_headline.py tests/test_qwen3_5.py -m gpu -x -q > $OUT/tests.log 2>&1; grep -n "passed\|failed" $OUT/tests.log; grep -n "^FAILED\|^ERROR" $OUT/tests.log | head
export BENCH_GREEDY=1
timeout 200 python tools/bench_engine.py qwen3-8b 256 128 128 8 128 2>&1 | grep "tok/s" | cut -c60-140
timeout 200 python tools/bench_engine.py qwen3-8b 256 32 128 8 128 2>&1 | grep "tok/s" | cut -c60-140
