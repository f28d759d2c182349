#!/bin/bash
OUT=gpurun_out/${1:-r3L}
mkdir -p $OUT
run() { echo "== $*"; env "$@" timeout 200 python tools/bench_vit_batch.py qwen3-vl-2b 1 2>&1 | tail -1; }
run X=1
run CM_KSPLIT_CAP=1
run CM_KSPLIT_CAP=256
run CM_KSPLIT_CAP=512
run CM_GEMM256=0
run CM_GEMM256=0 CM_KSPLIT_CAP=1
run CM_GEMM256_MIN_BLOCKS=56
run CM_GEMM_BM=64
