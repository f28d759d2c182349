#!/bin/bash
OUT=$1
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); p = d.get('parity') or {}
        print(d['value'], d['ms_per_step'], d['roofline_step']['frac'], p.get('logit_rel'), p.get('ok'))"; }
for e in 0 1 0 1; do
  echo -n "qwen3.5-0.8b CM_ATTN_FUSED_COMBINE=$e: "
  CM_ATTN_FUSED_COMBINE=$e timeout 300 python bench.py --model qwen3.5-0.8b --steps 64 --warmup 8 --no-cpu-baseline 2>$OUT/ab_err.log | line
  echo -n "qwen3-8b tp-local 8 rank 0 CM_ATTN_FUSED_COMBINE=$e: "
  CM_ATTN_FUSED_COMBINE=$e timeout 300 python bench.py --model qwen3-8b --tp-local 8 --rank 0 --steps 64 --warmup 8 --no-cpu-baseline 2>$OUT/ab_err.log | line
  echo -n "qwen3-8b launches (engine off) CM_ATTN_FUSED_COMBINE=$e: "
  CM_ATTN_FUSED_COMBINE=$e timeout 300 python bench.py --model qwen3-8b --engine -1 --steps 32 --warmup 4 --no-cpu-baseline 2>$OUT/ab_err.log | line
done
