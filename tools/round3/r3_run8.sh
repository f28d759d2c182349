#!/bin/bash
OUT=gpurun_out/${1:-r3m}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
kt() { local n=$1; shift
    timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt_$n -o $n -- "$@" > $OUT/kt_$n.log 2>&1
    python tools/rocpd_stats.py $(ls $OUT/kt_$n/*_results.db | head -1) $OUT/${n}_kernel_stats.csv > /dev/null 2>>$OUT/kt_$n.log
    rm -rf $OUT/kt_$n; }
kt batched_decode_b128 python tools/prof_batch.py 128
head -24 $OUT/batched_decode_b128_kernel_stats.csv
