#!/bin/bash
OUT=gpurun_out/${1:-r3y}
mkdir -p $OUT
timeout 900 python bench.py --model qwen3-vl-2b > $OUT/bench_vl.json 2> $OUT/bench_vl.err; tail -3 $OUT/bench_vl.err; cat $OUT/bench_vl.json
timeout 600 python -m pytest tests/test_gpu_parity_headline.py -m gpu -x -q -k large_decode > $OUT/tests.log 2>&1; tail -2 $OUT/tests.log
