#!/bin/bash
OUT=gpurun_out/${1:-r3o}
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity_headline.py tests/test_qwen3_5.py tests/test_gpu_engine.py -m gpu -x -q > $OUT/tests.log 2>&1; tail -5 $OUT/tests.log
