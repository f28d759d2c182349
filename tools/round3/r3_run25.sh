#!/bin/bash
OUT=gpurun_out/${1:-r3H}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_qwen3_5.py tests/test_qwen3_5_vl.py tests/test_gpu_tp_shards.py tests/test_gpu_kv_quant.py -m gpu -x -q > $OUT/tests.log 2>&1; grep -n "passed\|failed" $OUT/tests.log; grep -n "^FAILED\|^ERROR\|Error" $OUT/tests.log | head
for d in 1 0; do
  CM_GDN_DEFER_NORM=$d timeout 300 python bench.py --model qwen3.5-0.8b --no-cpu-baseline > $OUT/b08_$d.json 2>/dev/null
  CM_GDN_DEFER_NORM=$d timeout 600 python bench.py --model qwen3.8-27b --steps 32 --warmup 4 --no-cpu-baseline > $OUT/b27_$d.json 2>/dev/null
done
python - <<PY
import json
for n in ("b08_1","b08_0","b27_1","b27_0"):
    for l in open("$OUT/%s.json"%n):
        if l.startswith("{"):
            d=json.loads(l); print(n, d["value"], d["ms_per_step"], d["roofline_step"]["frac"])
PY
