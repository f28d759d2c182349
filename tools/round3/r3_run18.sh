#!/bin/bash
OUT=gpurun_out/${1:-r3A}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
kt() { local n=$1; shift
    timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt_$n -o $n -- "$@" > $OUT/kt_$n.log 2>&1
    python tools/rocpd_stats.py $(ls $OUT/kt_$n/*_results.db | head -1) $OUT/${n}_kernel_stats.csv > /dev/null 2>>$OUT/kt_$n.log
    rm -rf $OUT/kt_$n; }
kt vit1 python tools/bench_vit_batch.py qwen3-vl-2b 1
kt vit5 python tools/bench_vit_batch.py qwen3-vl-2b 5
head -16 $OUT/vit1_kernel_stats.csv | cut -c1-150
head -16 $OUT/vit5_kernel_stats.csv | cut -c1-150
