#!/bin/bash
OUT=gpurun_out/${1:-r3z}
mkdir -p $OUT
timeout 300 python tools/bench_vit_batch.py qwen3-vl-2b 1,2,4,5 > $OUT/vit_batch.log 2>&1; cat $OUT/vit_batch.log | tail -5
timeout 600 python -m pytest tests/test_qwen3_vl.py tests/test_qwen3_5_vl.py -m gpu -x -q > $OUT/tests.log 2>&1; tail -2 $OUT/tests.log
