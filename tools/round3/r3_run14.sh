#!/bin/bash
OUT=gpurun_out/${1:-r3v}
mkdir -p $OUT
export BENCH_GREEDY=1
for cfg in "1 0"; do
  set -- $cfg
  if [ "$2" = "0" ]; then unset CM_ATTN_MFMA_MIN; else export CM_ATTN_MFMA_MIN=$2; fi
  CM_ATTN_BATCH_NS_MIN=$1 timeout 120 python tools/bench_engine.py qwen3-8b 256 128 128 8 64,128 > $OUT/eng_$1_$2.log 2>&1
  echo "ns_min=$1 mfma_min=$2"; grep "tok/s" $OUT/eng_$1_$2.log | cut -c60-130
done
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_parity_qwen3.py tests/test_qwen3_5.py tests/test_gpu_kv_quant.py tests/test_gpu_parity_headline.py tests/test_gpu_tp_shards.py -m gpu -x -q > $OUT/tests.log 2>&1; grep -n "passed\|failed" $OUT/tests.log; grep -n "^FAILED\|^ERROR\|Error" $OUT/tests.log | head
