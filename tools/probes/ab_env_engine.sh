#!/bin/bash
# CM_ENG_TUNE variants of the persistent kernel, alternating:  bash tools/probes/ab_env_engine.sh <outdir> <model> <tune A> <tune B>
OUT=$1; M=$2; A=$3; B=$4
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print(d['value'], d['ms_per_step'], d['roofline'].get('us_per_launch'), d['roofline_step']['frac'])"; }
for t in $A $B $A $B $A $B; do
  echo -n "$M CM_ENG_TUNE=$t: "
  CM_ENG_TUNE=$t timeout 300 python bench.py --model $M --no-cpu-baseline --steps 64 --warmup 8 2>/dev/null | line
done
