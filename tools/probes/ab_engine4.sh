#!/bin/bash
OUT=$1
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print(d['value'], d['ms_per_step'], d['roofline'].get('us_per_launch'), d['roofline_step']['frac'])"; }
run() { local m=$1; shift
  echo -n "$m $*: "
  env "$@" timeout 300 python bench.py --model $m --no-cpu-baseline --steps 64 --warmup 8 2>$OUT/ab_err.log | line; }
for t in 0x880 0x881 0x880 0x881; do run qwen3-8b CM_ENG_TUNE=$t; done
for t in 0x0 0x1 0x0 0x1; do run qwen3-0.6b CM_ENG_TUNE=$t; done
