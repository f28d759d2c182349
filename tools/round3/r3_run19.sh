#!/bin/bash
OUT=gpurun_out/${1:-r3B}
mkdir -p $OUT
for ks in 1 2 4 8; do echo "CM_VIT_KSPLIT=$ks"; CM_VIT_KSPLIT=$ks timeout 300 python tools/bench_vit_batch.py qwen3-vl-2b 1,5 2>&1 | tail -2; done | tee $OUT/vit_ksplit.log
timeout 600 python -m pytest tests/test_qwen3_vl.py tests/test_qwen3_5_vl.py -m gpu -x -q > $OUT/tests.log 2>&1; tail -2 $OUT/tests.log
