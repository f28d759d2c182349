#!/bin/bash
OUT=gpurun_out/${1:-r3h}
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity_qwen3.py tests/test_gpu_parity_headline.py tests/test_qwen3_vl.py tests/test_qwen3_5_vl.py tests/test_golden_qwen3.py tests/test_gpu_kv_quant.py -q -m gpu --timeout 600 -p no:cacheprovider -x > $OUT/tests.log 2>&1; tail -5 $OUT/tests.log
timeout 200 python tools/bench_vision.py 5 qwen3-vl-2b > $OUT/vision_bench.json 2> $OUT/vision_bench.err; tail -1 $OUT/vision_bench.json
timeout 100 python tools/prefill_sweep.py 0 128,512,1024,2048 > $OUT/prefill_parity.log 2>&1; tail -4 $OUT/prefill_parity.log

cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
kt() { local n=$1; shift
    timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt_$n -o $n -- "$@" > $OUT/kt_$n.log 2>&1
    python tools/rocpd_stats.py $(ls $OUT/kt_$n/*_results.db | head -1) $OUT/${n}_kernel_stats.csv > /dev/null 2>>$OUT/kt_$n.log
    rm -rf $OUT/kt_$n; }
kt prefill_1024 python tools/prof_prefill.py qwen3-8b 1024
head -12 $OUT/prefill_1024_kernel_stats.csv
