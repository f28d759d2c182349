#!/bin/bash
# checkpoint of the round: whole -m gpu suite, the bench lines of every BASELINE config, the engine in both sampling modes
OUT=gpurun_out/${1:-r3D}
mkdir -p $OUT
timeout 1200 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider > $OUT/tests.log 2>&1; tail -4 $OUT/tests.log
timeout 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
for mdl in qwen3-0.6b qwen3.5-0.8b qwen3-vl-2b qwen3.8-27b; do
  timeout 600 python bench.py --model $mdl $([ $mdl = qwen3.8-27b ] && echo "--steps 32 --warmup 4") > $OUT/bench_$mdl.json 2> $OUT/bench_$mdl.err
done
timeout 300 python tools/bench_engine.py qwen3-8b 256 128 128 8 64,128 > $OUT/eng.log 2>&1; grep "tok/s" $OUT/eng.log | cut -c1-150
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        for l in open(f):
            if l.startswith("{"):
                d = json.loads(l); print(f.split("/")[-1], d["value"], d["ms_per_step"], d["roofline_step"]["frac"] if "frac" in d["roofline_step"] else "", d["roofline"].get("us_per_launch"), (d.get("parity") or {}).get("logit_rel"), (d.get("parity") or {}).get("ok"), (d.get("prefill") or {}).get("ms"), (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "ERR", e)
PY
