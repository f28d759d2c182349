#!/bin/bash
OUT=gpurun_out/${1:-r3q}
mkdir -p $OUT
timeout 200 python tools/tune_gemv27.py "-" > $OUT/sweep_nw5.log 2>&1; cat $OUT/sweep_nw5.log
CM_GEMV_NW5=0 timeout 200 python tools/tune_gemv27.py "-" > $OUT/sweep_nw4.log 2>&1; cat $OUT/sweep_nw4.log
timeout 600 python -m pytest tests/test_qwen3_5.py tests/test_gpu_tp_shards.py -m gpu -x -q > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
timeout 600 python bench.py --model qwen3.8-27b --steps 32 --warmup 4 --no-cpu-baseline > $OUT/bench27.json 2> $OUT/bench27.err; cut -c1-400 $OUT/bench27.json
