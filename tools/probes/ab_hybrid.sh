#!/bin/bash
OUT=$1
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); p = d.get('parity') or {}
        print(d['value'], d['ms_per_step'], d['roofline_step']['frac'], p.get('logit_rel'), p.get('ok'))"; }
for m in qwen3.5-0.8b qwen3.5-2b; do
for e in 0 1 0 1; do
  echo -n "$m CM_ENGINE_HYBRID=$e: "
  CM_ENGINE_HYBRID=$e timeout 300 python bench.py --model $m --steps 64 --warmup 8 --no-cpu-baseline 2>$OUT/ab_err.log | line
done; done
