#!/bin/bash
OUT=gpurun_out/${1:-r3C}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
pmc() { local n=$1 c=$2; shift 2
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$n -o $n -- "$@" > $OUT/pmc_$n.log 2>&1
    python tools/pmc_summary.py $OUT/pmc_$n cm:: > $OUT/pmc_$n.json 2>>$OUT/pmc_$n.log
    rm -rf $OUT/pmc_$n; }
export CM_VIT_KSPLIT=${2:-1}
pmc vit_a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES" python tools/bench_vit_batch.py qwen3-vl-2b 1
pmc vit_b "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE" python tools/bench_vit_batch.py qwen3-vl-2b 1
python - <<PY
import json
for n in ("vit_a","vit_b"):
    try:
        d=json.load(open("$OUT/pmc_%s.json"%n))
        for r in d:
            if "attn_prefill" in r["kernel"] or "gemm_bf16_kernel" in r["kernel"]:
                print(n, {k:(round(v,1) if isinstance(v,float) else v) for k,v in r.items()})
    except Exception as e: print(n, "ERR", e)
PY
tail -3 $OUT/pmc_vit_b.log
