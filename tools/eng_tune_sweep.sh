#!/bin/bash
# CM_ENG_TUNE sweep of the persistent decode kernel:  tools/eng_tune_sweep.sh <model> <tune values...>
M=${1:-qwen3-8b}; shift
for t in "$@"; do
  echo -n "$M CM_ENG_TUNE=$t: "; CM_ENG_TUNE=$t timeout 120 python bench.py --model $M --no-cpu-baseline --steps 128 --warmup 8 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], d['ms_per_step'], d['roofline'].get('us_per_launch'), d['roofline_step']['frac'])"
done
