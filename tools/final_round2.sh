#!/bin/bash
# Runs ON the GPU box (through gpurun): the whole -m gpu suite on the final build, then the 64-row GEMM tile A/B on the engine.
OUT=gpurun_out/${1:-fin2}
mkdir -p $OUT
timeout 300 python -m pytest tests -q -m gpu > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
BENCH_GREEDY=1 timeout 60 python tools/bench_engine.py qwen3-8b 128 128 128 8 32,64 > $OUT/eng_bm64.log 2>&1; grep "tok/s" $OUT/eng_bm64.log
CM_GEMM_BM=128 BENCH_GREEDY=1 timeout 60 python tools/bench_engine.py qwen3-8b 128 128 128 8 32,64 > $OUT/eng_bm128.log 2>&1; grep "tok/s" $OUT/eng_bm128.log
timeout 40 python tools/prefill_sweep.py 0 16,32,64,128 > $OUT/prefill_short.log 2>&1; grep prefill $OUT/prefill_short.log
