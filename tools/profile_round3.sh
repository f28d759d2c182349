#!/bin/bash
# Runs ON the GPU box (through gpurun): the round's rocprofv3 evidence + bench lines -> gpurun_out/$1/ (copied into profiles/ afterwards).
#   kernel-trace stats: Qwen3-8B decode (persistent kernel, launch path), 1024-token prefill, Qwen3-0.6B, Qwen3.5-0.8B, Qwen3.8-27B, ViT tower
#   PMC (separate passes, kernel-trace only): FETCH_SIZE, WRITE_SIZE on the 8B decode; MFMA counters on the prefill
#   bench lines: default (8B, with the CPU leg), 0.6B, 0.8B (CPU leg = the hybrid C port), 27B, vision
OUT=gpurun_out/${1:-r03p}
mkdir -p $OUT
git_rev=$(cat .git_rev 2>/dev/null || echo unknown)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
kt() { local n=$1; shift
    timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt_$n -o $n -- "$@" > $OUT/kt_$n.log 2>&1
    python tools/rocpd_stats.py $(ls $OUT/kt_$n/*_results.db | head -1) $OUT/${n}_kernel_stats.csv > /dev/null 2>>$OUT/kt_$n.log
    rm -rf $OUT/kt_$n; }
pmc() { local n=$1 c=$2; shift 2
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$n -o $n -- "$@" > $OUT/pmc_$n.log 2>&1
    python tools/pmc_summary.py $OUT/pmc_$n cm:: > $OUT/pmc_$n.json 2>>$OUT/pmc_$n.log
    rm -rf $OUT/pmc_$n; }
timeout 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
kt decode_qwen3_8b_engine python bench.py --no-cpu-baseline --steps 32 --warmup 4
kt decode_qwen3_8b_launches python bench.py --no-cpu-baseline --steps 32 --warmup 4 --engine -1
kt prefill_1024_qwen3_8b python tools/prof_prefill.py qwen3-8b 1024
pmc fetch_8b FETCH_SIZE python bench.py --no-cpu-baseline --steps 8 --warmup 2 --no-graph
pmc write_8b WRITE_SIZE python bench.py --no-cpu-baseline --steps 8 --warmup 2 --no-graph
pmc mfma_prefill_8b "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" python tools/prof_prefill.py qwen3-8b 1024
python tools/merge_traffic.py $OUT/pmc_fetch_8b.json $OUT/pmc_write_8b.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE -- python bench.py --no-cpu-baseline --steps 8 --warmup 2 --no-graph (tools/profile_round3.sh, source tree $git_rev)" > $OUT/pmc_traffic_decode.json 2>/dev/null
timeout 200 python bench.py --model qwen3-0.6b > $OUT/bench_qwen3_0_6b.json 2> $OUT/bench_qwen3_0_6b.err
timeout 300 python bench.py --model qwen3.5-0.8b > $OUT/bench_qwen3_5_0_8b.json 2> $OUT/bench_qwen3_5_0_8b.err
timeout 300 python bench.py --model qwen3.8-27b --steps 32 --warmup 4 > $OUT/bench_qwen3_8_27b.json 2> $OUT/bench_qwen3_8_27b.err
kt decode_qwen3_5_0p8b python bench.py --no-cpu-baseline --model qwen3.5-0.8b --steps 64 --warmup 4
kt vit_tower_24x1024_784 python tools/bench_vision.py 5
python tools/bench_vision.py 5 > $OUT/vision_bench.json 2>/dev/null
ls -la $OUT | head -40
tail -c 600 $OUT/bench_qwen3_5_0_8b.json; tail -c 300 $OUT/bench_qwen3_8_27b.json; tail -c 300 $OUT/bench_qwen3_0_6b.json
