# per-shape durations of the int8 prompt GEMM (grid = weight tiles x m-panels x K slices): qkv / o / gate|up / down of a 1024-token pass
OUT=$1
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/kt_q8g -o q8g -- python tools/bench_q8_prefill.py qwen3-8b q8_0 1 1024 > $OUT/kt_q8g.log 2>&1
ROCPD_BY_GRID=gemm_q8 python tools/rocpd_stats.py $(ls $OUT/kt_q8g/*_results.db | head -1) | grep "grid="
rm -rf $OUT/kt_q8g
