#!/bin/bash
export ROUND_REPS=5
for n in 64 32 16; do
  for e in "CM_GEMM256_WST=3" "X=1" "CM_GEMM256_WST=3" "X=1"; do
    echo -n "$e: "; env $e python tools/probes/round_profile.py $n 192 2>&1 | grep "round of"
  done
done
