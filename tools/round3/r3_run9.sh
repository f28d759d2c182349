#!/bin/bash
OUT=gpurun_out/${1:-r3n}
mkdir -p $OUT
timeout 200 python tools/bench_gemm.py 128 0 3 > $OUT/gemm_128_parity.log 2>&1; tail -4 $OUT/gemm_128_parity.log
timeout 200 python tools/bench_gemm.py 96 0 3 > $OUT/gemm_96_parity.log 2>&1; tail -4 $OUT/gemm_96_parity.log
timeout 200 python tools/bench_gemm.py 256 0 3 > $OUT/gemm_256_parity.log 2>&1; tail -4 $OUT/gemm_256_parity.log
BENCH_GREEDY=1 timeout 200 python tools/bench_engine.py qwen3-8b 256 128 128 8 32,64,128 > $OUT/eng_g256.log 2>&1; grep "tok/s" $OUT/eng_g256.log
CM_GEMM256_MIN_M=512 BENCH_GREEDY=1 timeout 200 python tools/bench_engine.py qwen3-8b 256 128 128 8 32,64,128 > $OUT/eng_g128.log 2>&1; grep "tok/s" $OUT/eng_g128.log
