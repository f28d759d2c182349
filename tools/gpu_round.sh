#!/bin/bash
# The ONE GPU-box script (runs through gpurun from the repo root):
#     gpurun --timeout T -- 'bash tools/gpu_round.sh <outdir> <stage> [<stage> ...]'
# Every stage writes under gpurun_out/<outdir>/ (merged back by gpurun; what is to be judged is copied into profiles/ afterwards).
# Stages (run in the order given; a stage argument is attached with ':' and may be a comma list):
#   tests[:<pytest -k expr>]     python -m pytest tests -m gpu               -> tests.log
#   smoke                        __graft_entry__.smoke()                      -> smoke.log
#   bench[:m1,m2,...]            bench.py of each model (default qwen3-8b)    -> bench_<model>.json / .err
#   benchx:<name>:<args...>      bench.py with extra args ('+' separates args) -> bench_<name>.json
#   tplocal[:model]              per-rank shard timing, TP = 2 / 4 / 8 rank 0 (+ last rank at 8), + the 1-rank FORCE_RCCL run
#   kt:<name>:<cmd...>           rocprofv3 --kernel-trace --stats of a command ('+' separates words) -> <name>_kernel_stats.csv
#   ktbench[:m1,...]             kt of bench.py --no-cpu-baseline for each model
#   traffic[:m1,...]             FETCH_SIZE / WRITE_SIZE pmc passes (separate runs) of bench.py for each model -> pmc_traffic_decode_<model>.json
#   pmc:<name>:<counters>:<cmd>  one --pmc pass ('+' separates counters and words)  -> pmc_<name>.json
#   engine[:args]                tools/bench_engine.py qwen3-8b 256 128 128 8 32,64,128  (or '+'-separated args) -> engine.log
#   py:<name>:<script+args>      python <script> ...                          -> <name>.log
#   sh:<name>:<file>             bash <file> <outdir>                          -> <name>.log
OUT=gpurun_out/${1:-round}; shift
mkdir -p $OUT
REV=$(cat .git_rev 2>/dev/null || echo unknown)
export TMPDIR=/tmp
san() { echo "$1" | tr '.-' '__'; }
kt() { local n=$1; shift
    ( cd /tmp; cd $GRAFT_REPO_ROOT; timeout 420 rocprofv3 --kernel-trace --stats -d $OUT/kt_$n -o $n -- "$@" > $OUT/kt_$n.log 2>&1 )
    python tools/rocpd_stats.py $(ls $OUT/kt_$n/*_results.db 2>/dev/null | head -1) $OUT/${n}_kernel_stats.csv > /dev/null 2>>$OUT/kt_$n.log
    rm -rf $OUT/kt_$n; }
pmc() { local n=$1 c=$2; shift 2
    timeout 420 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$n -o $n -- "$@" > $OUT/pmc_$n.log 2>&1
    python tools/pmc_summary.py $OUT/pmc_$n cm:: > $OUT/pmc_$n.json 2>>$OUT/pmc_$n.log
    rm -rf $OUT/pmc_$n; }
bargs() { case $1 in qwen3.8-27b) echo "--steps 32 --warmup 4";; *) echo "";; esac; }
for st in "$@"; do
  IFS=':' read -r kind a1 a2 a3 <<< "$st"
  echo "=== stage $st ($(date +%T))"
  case $kind in
    tests) timeout 1500 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider ${a1:+-k "$a1"} > $OUT/tests.log 2>&1
           grep -n "passed\|failed" $OUT/tests.log | tail -2; grep -n "^FAILED\|^ERROR" $OUT/tests.log | head -20;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log;;
    bench) for m in $(echo ${a1:-qwen3-8b} | tr ',' ' '); do
             timeout 700 python bench.py --model $m $(bargs $m) > $OUT/bench_$(san $m).json 2> $OUT/bench_$(san $m).err; done;;
    benchx) timeout 700 python bench.py $(echo "$a2" | tr '+' ' ') > $OUT/bench_$a1.json 2> $OUT/bench_$a1.err;;
    tplocal) m=${a1:-qwen3-8b}
           for nr in "2 0" "4 0" "8 0" "8 7"; do set -- $nr
             timeout 300 python bench.py --model $m --tp-local $1 --rank $2 $(bargs $m) > $OUT/bench_$(san $m)_tp$1_r$2.json 2> $OUT/bench_$(san $m)_tp$1_r$2.err; done
           timeout 300 python bench.py --model $m --force-rccl --no-cpu-baseline $(bargs $m) > $OUT/bench_$(san $m)_force_rccl.json 2> $OUT/bench_$(san $m)_force_rccl.err
           timeout 300 python bench.py --model $m --engine -1 --no-cpu-baseline $(bargs $m) > $OUT/bench_$(san $m)_launches.json 2> $OUT/bench_$(san $m)_launches.err;;
    kt) kt $a1 $(echo "$a2" | tr '+' ' ');;
    ktbench) for m in $(echo ${a1:-qwen3-8b} | tr ',' ' '); do kt decode_$(san $m) python bench.py --model $m --no-cpu-baseline --steps 32 --warmup 4; done;;
    traffic) for m in $(echo ${a1:-qwen3-8b} | tr ',' ' '); do s=$(san $m)
             pmc fetch_$s FETCH_SIZE python bench.py --model $m --no-cpu-baseline --steps 8 --warmup 2 --no-graph
             pmc write_$s WRITE_SIZE python bench.py --model $m --no-cpu-baseline --steps 8 --warmup 2 --no-graph
             python tools/merge_traffic.py $OUT/pmc_fetch_$s.json $OUT/pmc_write_$s.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python bench.py --model $m --no-cpu-baseline --steps 8 --warmup 2 --no-graph (tools/gpu_round.sh traffic, source tree $REV)" > $OUT/pmc_traffic_decode_$s.json 2>/dev/null; done;;
    pmc) pmc $a1 "$(echo "$a2" | tr '+' ' ')" $(echo "$a3" | tr '+' ' ');;
    engine) timeout 400 python tools/bench_engine.py $(echo "${a1:-qwen3-8b+256+128+128+8+32,64,128}" | tr '+' ' ') > $OUT/engine.log 2>&1; grep "tok/s" $OUT/engine.log | cut -c1-160;;
    py) timeout 900 python $(echo "$a2" | tr '+' ' ') > $OUT/$a1.log 2>&1; tail -5 $OUT/$a1.log | cut -c1-300;;
    sh) timeout 1800 bash $a2 $OUT > $OUT/$a1.log 2>&1; tail -5 $OUT/$a1.log | cut -c1-300;;
    *) echo "unknown stage $st";;
  esac
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        for l in open(f):
            if l.startswith("{"):
                d = json.loads(l); p = d.get("parity") or {}
                lk = [k for k in p if k.startswith("model_written_cache_ctx")]
                print(f.split("/")[-1], d["value"], d["ms_per_step"], (d["roofline_step"] or {}).get("frac"), (d.get("roofline") or {}).get("us_per_launch"),
                      p.get("logit_rel"), p.get("ok"), (p.get(lk[0]) if lk else None), (d.get("prefill") or {}).get("ms"), (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "ERR", e)
PY
