#!/bin/bash
# GPU box: NREP=2 persistent kernel + ViT changes -- targeted tests, the vision bench, per-kernel times of the 1024-token prefill
OUT=gpurun_out/${1:-r3b}
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_engine_chain.py tests/test_qwen3_vl.py tests/test_qwen3_5_vl.py -q -m gpu --timeout 600 -p no:cacheprovider > $OUT/tests.log 2>&1; tail -8 $OUT/tests.log
timeout 200 python tools/bench_vision.py 5 qwen3-vl-2b > $OUT/vision_bench.json 2> $OUT/vision_bench.err; tail -2 $OUT/vision_bench.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
kt() { local n=$1; shift
    timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt_$n -o $n -- "$@" > $OUT/kt_$n.log 2>&1
    python tools/rocpd_stats.py $(ls $OUT/kt_$n/*_results.db | head -1) $OUT/${n}_kernel_stats.csv > /dev/null 2>>$OUT/kt_$n.log
    rm -rf $OUT/kt_$n; }
kt prefill_1024 python tools/prof_prefill.py qwen3-8b 1024
kt vit_tower python tools/bench_vision.py 3
head -14 $OUT/prefill_1024_kernel_stats.csv; head -12 $OUT/vit_tower_kernel_stats.csv
