#!/bin/bash
OUT=gpurun_out/${1:-r3c}
mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_parity_headline.py -q -m gpu --timeout 300 -p no:cacheprovider -k "wide_gemm or prefill_1024" > $OUT/tests.log 2>&1; tail -8 $OUT/tests.log
timeout 200 python tools/bench_gemm.py 1024 0 3 > $OUT/gemm_1024_parity.log 2>&1; cat $OUT/gemm_1024_parity.log | tail -5
timeout 200 python tools/bench_gemm.py 1024 1 3 > $OUT/gemm_1024_plain.log 2>&1; cat $OUT/gemm_1024_plain.log | tail -5
timeout 200 python tools/bench_gemm.py 2048 0 3 > $OUT/gemm_2048_parity.log 2>&1; cat $OUT/gemm_2048_parity.log | tail -5
timeout 100 python -m pytest tests/test_gpu_engine_chain.py -q -m gpu --timeout 300 -p no:cacheprovider > $OUT/tests_chain.log 2>&1; tail -3 $OUT/tests_chain.log
