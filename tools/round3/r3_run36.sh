#!/bin/bash
OUT=gpurun_out/${1:-r3S}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_qwen3_5.py tests/test_qwen3_5_vl.py tests/test_gpu_tp_shards.py tests/test_gpu_ref_kernels.py -m gpu -x -q > $OUT/tests.log 2>&1; grep -n "passed\|failed" $OUT/tests.log; grep -n "^FAILED\|^ERROR" $OUT/tests.log | head
for mdl in qwen3.5-0.8b qwen3.8-27b; do timeout 600 python bench.py --model $mdl --steps 16 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$mdl', d['value'], d['prefill'])"; done
