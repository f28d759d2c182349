#!/bin/bash
# A/B of the decode-round launch fusions (CM_ATTN_OUT1: single-split attention writes its own output / planes; CM_GEMM_NORM_FUSED:
# RMSNorm on the split-K reduction launch): correctness subset first, then the serving loop both ways
OUT=$1
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -x -k "large_decode_groups or engine or batch or tp_group" 2>&1 | tail -6
for f in 0 1; do
  echo "--- CM_ATTN_OUT1=$f CM_GEMM_NORM_FUSED=$f"
  CM_ATTN_OUT1=$f CM_GEMM_NORM_FUSED=$f timeout 300 python tools/bench_engine.py qwen3-8b 256 128 128 8 64,128 2>&1 | grep "tok/s" | cut -c1-170
done
