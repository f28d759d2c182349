#!/bin/bash
OUT=gpurun_out/${1:-r3P}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_qwen3_5.py tests/test_gpu_engine_chain.py -m gpu -x -q > $OUT/tests.log 2>&1; grep -n "passed\|failed" $OUT/tests.log; grep -n "^FAILED\|^ERROR\|Error\|assert" $OUT/tests.log | head
for e in 0 -1; do timeout 600 python bench.py --model qwen3.8-27b --engine $e --steps 32 --warmup 4 --no-cpu-baseline 2>$OUT/b27_$e.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('engine=$e', d['value'], d['ms_per_step'], d['roofline_step']['frac'], d['roofline'].get('us_per_launch'), d['roofline'].get('frac'), d['config']['decode_path'])"; tail -2 $OUT/b27_$e.err | grep -v amdgpu; done
