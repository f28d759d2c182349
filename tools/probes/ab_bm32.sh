#!/bin/bash
export ROUND_REPS=5
for n in 32 16 8 48; do
  for e in "X=1" "CM_GEMM256_MIN_M=9" "CM_GEMM256_MIN_M=9 CM_GEMM256_BM32=0" "X=1" "CM_GEMM256_MIN_M=9" "CM_GEMM256_MIN_M=9 CM_GEMM256_BM32=0"; do
    echo -n "$e: "; env $e python tools/probes/round_profile.py $n 192 2>&1 | grep "round of"
  done
done
