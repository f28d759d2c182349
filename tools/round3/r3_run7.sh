#!/bin/bash
OUT=gpurun_out/${1:-r3l}
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_engine.py tests/test_gpu_parity_qwen3.py tests/test_qwen3_5.py tests/test_gpu_tp_shards.py -q -m gpu --timeout 600 -p no:cacheprovider > $OUT/tests.log 2>&1; grep -E "passed|failed" $OUT/tests.log; grep -E "^FAILED|^ERROR|^E " $OUT/tests.log | head -10
BENCH_GREEDY=1 timeout 200 python tools/bench_engine.py qwen3-8b 256 128 128 8 32,64,128 > $OUT/eng_head_gemm.log 2>&1; grep "tok/s" $OUT/eng_head_gemm.log
CM_LM_HEAD_GEMM_MIN=0 BENCH_GREEDY=1 timeout 200 python tools/bench_engine.py qwen3-8b 256 128 128 8 32,64,128 > $OUT/eng_head_gemv.log 2>&1; grep "tok/s" $OUT/eng_head_gemv.log
