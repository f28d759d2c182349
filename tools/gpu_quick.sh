#!/bin/bash
# Runs ON the GPU box (through gpurun): quick loop for the persistent decode kernel -- its parity tests, one traced launch,
# and the headline bench for each CM_ENG_CFG given.   usage: tools/gpu_quick.sh <outdir> [cfg ...]
OUT=gpurun_out/${1:-q}; shift
mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_engine_chain.py tests/test_gpu_parity_headline.py -x -q -m gpu > $OUT/tests.log 2>&1
tail -3 $OUT/tests.log
CFGS=${@:-4,4}
first=1
for c in $CFGS; do
    if [ $first = 1 ]; then CM_ENG_CFG=$c timeout 120 python tools/engine_trace.py qwen3-8b 4 > $OUT/trace_$c.log 2>&1; first=0; fi
    CM_ENG_CFG=$c timeout 120 python bench.py --no-cpu-baseline --steps 64 --warmup 8 > $OUT/bench_$c.json 2>$OUT/bench_$c.err
    python - <<PY
import json
for l in open("$OUT/bench_$c.json"):
    if l.startswith("{"):
        d = json.loads(l); print("$c", d["value"], d["ms_per_step"], d["roofline_step"]["frac"])
PY
done
