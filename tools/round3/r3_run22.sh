#!/bin/bash
OUT=gpurun_out/${1:-r3E}
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_engine.py tests/test_gpu_sampler.py -m gpu -x -q > $OUT/tests.log 2>&1; tail -5 $OUT/tests.log
timeout 300 python tools/bench_engine.py qwen3-8b 256 128 128 8 64,128 > $OUT/eng.log 2>&1; grep "tok/s" $OUT/eng.log | cut -c1-150
