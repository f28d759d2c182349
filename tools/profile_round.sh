#!/bin/bash
# Runs ON the GPU box (through gpurun): the round's rocprofv3 evidence -> gpurun_out/$1/ (copied into profiles/ afterwards).
#   kernel-trace stats: Qwen3-8B decode (persistent kernel and launch path), Qwen3.5-0.8B, Qwen3.8-27B, ViT tower
#   PMC (separate passes, kernel-trace only): FETCH_SIZE, WRITE_SIZE on the 8B and 0.8B decode; MFMA counters on the prefill
OUT=gpurun_out/${1:-r02p}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
kt() { # name, command...
    local n=$1; shift
    timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt_$n -o $n -- "$@" > $OUT/kt_$n.log 2>&1
    python tools/rocpd_stats.py $(ls $OUT/kt_$n/*_results.db | head -1) $OUT/${n}_kernel_stats.csv > /dev/null 2>>$OUT/kt_$n.log
    rm -rf $OUT/kt_$n
}
pmc() { # name, counters, command...
    local n=$1 c=$2; shift 2
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$n -o $n -- "$@" > $OUT/pmc_$n.log 2>&1
    python tools/pmc_summary.py $OUT/pmc_$n cm:: > $OUT/pmc_$n.json 2>>$OUT/pmc_$n.log
    rm -rf $OUT/pmc_$n
}
kt decode_qwen3_8b_engine python bench.py --no-cpu-baseline --steps 32 --warmup 4
kt decode_qwen3_8b_launches python bench.py --no-cpu-baseline --steps 32 --warmup 4 --engine -1
kt decode_qwen3_5_0p8b python bench.py --no-cpu-baseline --model qwen3.5-0.8b --steps 64 --warmup 4
kt decode_qwen3_8_27b python bench.py --no-cpu-baseline --model qwen3.8-27b --steps 16 --warmup 2
kt vit_tower_24x1024_784 python tools/bench_vision.py 5
pmc fetch_8b FETCH_SIZE python bench.py --no-cpu-baseline --steps 8 --warmup 2 --no-graph
pmc write_8b WRITE_SIZE python bench.py --no-cpu-baseline --steps 8 --warmup 2 --no-graph
pmc fetch_08b FETCH_SIZE python bench.py --no-cpu-baseline --model qwen3.5-0.8b --steps 8 --warmup 2 --no-graph
pmc write_08b WRITE_SIZE python bench.py --no-cpu-baseline --model qwen3.5-0.8b --steps 8 --warmup 2 --no-graph
pmc mfma_prefill_8b "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" python tools/prof_prefill.py qwen3-8b 1024
pmc mfma_vit "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" python tools/bench_vision.py 2
python tools/bench_vision.py 5 > $OUT/vision_bench.json 2>/dev/null
ls -la $OUT
