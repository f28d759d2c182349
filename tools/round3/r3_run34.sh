#!/bin/bash
OUT=gpurun_out/${1:-r3Q}
mkdir -p $OUT
run() { local mdl=$1; shift; echo -n "$mdl $* : "; env "$@" timeout 400 python bench.py --model $mdl --steps 32 --warmup 4 --no-cpu-baseline 2>/tmp/err.log | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['roofline_step']['frac'], d['roofline'].get('us_per_launch'))"; tail -1 /tmp/err.log | grep -v amdgpu.ids; }
run qwen3.8-27b CM_ENG_PF1024=8
run qwen3.8-27b CM_ENG_PF1024=6
run qwen3.8-27b CM_ENG_PF1024=8 CM_ENG_TUNE=0x880
run qwen3.8-27b CM_ENG_PF1024=8 CM_ENG_GBLK=1,1,3,1
run qwen3-0.6b CM_ENG_PF1024=8
run qwen3-0.6b CM_ENG_PF1024=6
run qwen3-0.6b CM_ENG_PF1024=4
