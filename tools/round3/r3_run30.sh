#!/bin/bash
OUT=gpurun_out/${1:-r3M}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_engine_chain.py -m gpu -x -q > $OUT/tests.log 2>&1; tail -15 $OUT/tests.log
for e in 0 -1; do timeout 300 python bench.py --model qwen3-0.6b --engine $e --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('engine=$e', d['value'], d['ms_per_step'], d['roofline_step']['frac'], d['config']['decode_path'])"; done
