#!/bin/bash
OUT=gpurun_out/${1:-r3R}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
kt() { local n=$1; shift
    timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt_$n -o $n -- "$@" > $OUT/kt_$n.log 2>&1
    python tools/rocpd_stats.py $(ls $OUT/kt_$n/*_results.db | head -1) $OUT/${n}_kernel_stats.csv > /dev/null 2>>$OUT/kt_$n.log
    rm -rf $OUT/kt_$n; }
for ctx in 8192 32768; do
timeout 300 python bench.py --ctx $ctx --steps 32 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('ctx $ctx', d['value'], d['ms_per_step'], d['roofline_step']['frac'], d['roofline_step']['bytes_per_token_per_gpu'], d['config']['decode_path'])"
done
kt decode_32k python bench.py --ctx 32768 --steps 16 --warmup 2 --no-cpu-baseline
head -8 $OUT/decode_32k_kernel_stats.csv | cut -c1-160
