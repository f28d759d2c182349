#!/bin/bash
OUT=gpurun_out/${1:-r3x}
mkdir -p $OUT
export BENCH_GREEDY=1
for mm in 65 33 17; do
  for hm in 33 17; do
  CM_GEMM256_MIN_M=$mm CM_LM_HEAD_GEMM_MIN=$hm timeout 200 python tools/bench_engine.py qwen3-8b 256 128 128 8 16,32,64 > $OUT/eng_$mm_$hm.log 2>&1
  echo "gemm256_min_m=$mm lm_head_gemm_min=$hm"; grep "tok/s" $OUT/eng_$mm_$hm.log | cut -c60-130
  done
done
