#!/bin/bash
OUT=gpurun_out/${1:-r3j}
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_engine.py tests/test_qwen3_5.py tests/test_gpu_parity_qwen3.py -q -m gpu --timeout 600 -p no:cacheprovider > $OUT/tests.log 2>&1; grep -E "passed|failed" $OUT/tests.log; grep -E "^FAILED|^ERROR|Error" $OUT/tests.log | head -10
BENCH_GREEDY=1 timeout 200 python tools/bench_engine.py qwen3-8b 256 128 128 8 64,128 > $OUT/eng_batchprefill.log 2>&1; grep "tok/s" $OUT/eng_batchprefill.log
BENCH_BATCH_PREFILL=0 BENCH_GREEDY=1 timeout 200 python tools/bench_engine.py qwen3-8b 256 128 128 8 64,128 > $OUT/eng_single.log 2>&1; grep "tok/s" $OUT/eng_single.log
