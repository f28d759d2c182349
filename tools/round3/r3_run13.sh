#!/bin/bash
OUT=gpurun_out/${1:-r3u}
mkdir -p $OUT
BENCH_GREEDY=1 timeout 300 python tools/bench_engine.py qwen3-8b 256 128 128 8 32,64,128 > $OUT/eng.log 2>&1; grep "tok/s" $OUT/eng.log
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_parity_qwen3.py tests/test_qwen3_5.py tests/test_gpu_kv_quant.py -m gpu -x -q > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
