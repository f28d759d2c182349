#!/bin/bash
# Runs ON the GPU box (through gpurun): end-of-round check -> gpurun_out/$1/: the whole -m gpu suite, the default bench line,
# and the batched-decode evidence at 64 sequences (kernel-trace stats + a FETCH_SIZE pass of the same command).
OUT=gpurun_out/${1:-fin}
mkdir -p $OUT
timeout 400 python -m pytest tests -q -m gpu > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
timeout 200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 600 $OUT/bench_default.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 150 rocprofv3 --kernel-trace --stats -d $OUT/kt_b64 -o b64 -- python tools/prof_batch.py 64 > $OUT/kt_b64.log 2>&1
python tools/rocpd_stats.py $(ls $OUT/kt_b64/*_results.db | head -1) $OUT/batched_decode_b64_kernel_stats.csv > /dev/null 2>>$OUT/kt_b64.log
rm -rf $OUT/kt_b64
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_b64 -o b64 -- python tools/prof_batch.py 64 > $OUT/pmc_b64.log 2>&1
python tools/pmc_summary.py $OUT/pmc_b64 gemvm > $OUT/pmc_fetch_batched_b64.json 2>>$OUT/pmc_b64.log
rm -rf $OUT/pmc_b64
ls -la $OUT
