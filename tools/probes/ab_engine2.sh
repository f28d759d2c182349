#!/bin/bash
OUT=$1
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); p = d.get('parity') or {}
        print(d['value'], d['ms_per_step'], d['roofline'].get('us_per_launch'), d['roofline_step']['frac'])"; }
run() { local lib=$1 m=$2; shift 2
  cp crane_amd/lib_$lib.bin crane_amd/libcrane_mi355.so
  echo -n "$lib $m $*: "
  env "$@" timeout 300 python bench.py --model $m --no-cpu-baseline --steps 64 --warmup 8 2>$OUT/ab_err.log | line; }
timeout 120 tools/probes/hop_probe
for rep in 1 2; do
  run base qwen3-8b X=1
  run new qwen3-8b X=1
  run new qwen3-8b CM_ENG_TUNE=0x10880
  run new qwen3-8b CM_ENG_TUNE=0x20880
done
run new qwen3-0.6b X=1
run new qwen3-0.6b CM_ENG_TUNE=0x10000
run new qwen3-vl-2b X=1
run new qwen3-vl-2b CM_ENG_TUNE=0x10880
cp crane_amd/lib_new.bin crane_amd/libcrane_mi355.so
