# PMC of the prompt GEMMs at 1024 rows, parity mode: kernels_gemmw4.hip (full, and with the loop's DMA + fragment reads + barrier ablated)
OUT=$1
export TMPDIR=/tmp
for d in 0 7; do
  CM_GEMMW4_DBG=$d timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_w4_$d -o w4 -- python tools/bench_gemm.py 1024 0 1 > $OUT/pmc_w4_$d.log 2>&1
  python tools/pmc_summary.py $OUT/pmc_w4_$d gemmw4 > $OUT/pmc_w4_dbg$d.json
  python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/pmc_w4_$d/**/*kernel_trace.csv", recursive=True)
if f:
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "gemmw4" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:60], r.get("Grid_Size_X") or r.get("Grid_Size"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in acc.items(): print("dbg $d", k, len(v), "avg us", sum(v) / len(v))
PY
  rm -rf $OUT/pmc_w4_$d
done
