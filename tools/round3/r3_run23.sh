#!/bin/bash
OUT=gpurun_out/${1:-r3F}
mkdir -p $OUT
for mdl in qwen3-0.6b qwen3.5-0.8b; do
for hm in 0 100000; do
  CM_ATTN_HEADS_MAX=$hm timeout 200 python bench.py --model $mdl --no-cpu-baseline > $OUT/b_${mdl}_$hm.json 2>/dev/null
  python - <<PY
import json
for l in open("$OUT/b_${mdl}_$hm.json"):
    if l.startswith("{"):
        d=json.loads(l); print("$mdl heads_max=$hm", d["value"], d["ms_per_step"])
PY
done; done
