#!/bin/bash
OUT=gpurun_out/${1:-r3t}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
kt() { local n=$1; shift
    timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt_$n -o $n -- "$@" > $OUT/kt_$n.log 2>&1
    python tools/rocpd_stats.py $(ls $OUT/kt_$n/*_results.db | head -1) $OUT/${n}_kernel_stats.csv > /dev/null 2>>$OUT/kt_$n.log
    rm -rf $OUT/kt_$n; }
export BENCH_GREEDY=1
kt engine_mr128 python tools/bench_engine.py qwen3-8b 256 128 128 8 128
grep "tok/s" $OUT/kt_engine_mr128.log
head -40 $OUT/engine_mr128_kernel_stats.csv
