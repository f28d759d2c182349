#!/bin/bash
# A/B of an environment switch on bench.py (same box, alternating):  tools/ab_env.sh <outdir> VAR v1 v2 [bench args...]
OUT=$1; VAR=$2; A=$3; B=$4; shift 4
for v in $A $B $A $B; do
  echo -n "$VAR=$v: "
  env $VAR=$v timeout 300 python bench.py --no-cpu-baseline --steps 64 --warmup 8 "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); p = d.get('parity') or {}
        print(d['value'], d['ms_per_step'], d['roofline'].get('us_per_launch'), d['roofline'].get('frac'), d['roofline_step']['frac'], p.get('logit_rel'), p.get('ok'))"
done
