#!/bin/bash
# Runs ON the GPU box (through gpurun): the whole -m gpu suite, then the headline bench (f16 pages, the default) and the bf16-page A/B.
#   usage: tools/r3_run.sh <outdir> [tests|bench|all]
OUT=gpurun_out/${1:-r3a}; WHAT=${2:-all}
mkdir -p $OUT
if [ $WHAT = all ] || [ $WHAT = tests ]; then
  timeout 900 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider > $OUT/tests.log 2>&1; tail -25 $OUT/tests.log
fi
if [ $WHAT = all ] || [ $WHAT = bench ]; then
  timeout 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 3000 $OUT/bench_default.json
  timeout 200 python bench.py --kv bf16 --no-cpu-baseline --steps 64 --warmup 8 > $OUT/bench_bf16.json 2> $OUT/bench_bf16.err
  timeout 200 python bench.py --no-cpu-baseline --steps 64 --warmup 8 > $OUT/bench_f16.json 2> $OUT/bench_f16.err
  python - <<PY
import json
for n in ("bench_default", "bench_bf16", "bench_f16"):
    try:
        for l in open("$OUT/%s.json" % n):
            if l.startswith("{"):
                d = json.loads(l); print(n, d["value"], d["ms_per_step"], d["roofline_step"]["frac"], d["roofline"].get("us_per_launch"), d.get("parity"))
    except Exception as e:
        print(n, "ERR", e)
PY
fi
