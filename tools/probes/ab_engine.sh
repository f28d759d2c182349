#!/bin/bash
# A/B of two builds of the library (crane_amd/lib_base.bin / lib_new.bin) on the persistent decode kernel: bench.py lines, alternating
OUT=$1
cp crane_amd/libcrane_mi355.so /tmp/lib_orig.so
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); p = d.get('parity') or {}
        lk = [k for k in p if k.startswith('model_written_cache_ctx')]
        print(d['value'], d['ms_per_step'], d['roofline'].get('us_per_launch'), d['roofline_step']['frac'], p.get('logit_rel'), p.get('ok'), (p.get(lk[0]) if lk else None))"; }
run() { # lib model env...
  local lib=$1 m=$2; shift 2
  cp crane_amd/lib_$lib.bin crane_amd/libcrane_mi355.so
  echo -n "$lib $m $*: "
  env "$@" timeout 300 python bench.py --model $m --no-cpu-baseline --steps 64 --warmup 8 2>$OUT/ab_err.log | line
}
for rep in 1 2; do
  run base qwen3-8b X=1
  run new qwen3-8b X=1
  run new qwen3-8b CM_ENG_GBLK=2,0,0,0
done
for m in qwen3-0.6b qwen3-vl-2b; do
  run base $m X=1
  run new $m X=1
  run base $m X=1
  run new $m X=1
done
cp crane_amd/lib_new.bin crane_amd/libcrane_mi355.so
