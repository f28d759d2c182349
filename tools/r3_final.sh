#!/bin/bash
# Last GPU action of the round: the whole -m gpu suite on HEAD, smoke(), the bench lines of all five BASELINE configs, the PMC
# traffic passes of the 0.6B persistent kernel, the engine in both sampling modes.   usage: tools/r3_final.sh <outdir>
OUT=gpurun_out/${1:-r3final}
mkdir -p $OUT
git_rev=$(cat .git_rev 2>/dev/null || echo unknown)
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider > $OUT/tests.log 2>&1; grep -n "passed\|failed" $OUT/tests.log; grep -n "^FAILED\|^ERROR" $OUT/tests.log | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
pmc() { local n=$1 c=$2; shift 2
    timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$n -o $n -- "$@" > $OUT/pmc_$n.log 2>&1
    python tools/pmc_summary.py $OUT/pmc_$n cm:: > $OUT/pmc_$n.json 2>>$OUT/pmc_$n.log
    rm -rf $OUT/pmc_$n; }
pmc fetch_qwen3_0_6b FETCH_SIZE python bench.py --model qwen3-0.6b --no-cpu-baseline --steps 8 --warmup 2 --no-graph
pmc write_qwen3_0_6b WRITE_SIZE python bench.py --model qwen3-0.6b --no-cpu-baseline --steps 8 --warmup 2 --no-graph
python tools/merge_traffic.py $OUT/pmc_fetch_qwen3_0_6b.json $OUT/pmc_write_qwen3_0_6b.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python bench.py --model qwen3-0.6b --no-cpu-baseline --steps 8 --warmup 2 --no-graph (tools/r3_final.sh, source tree $git_rev)" > $OUT/pmc_traffic_decode_qwen3_0_6b.json 2>/dev/null
cp $OUT/pmc_traffic_decode_qwen3_0_6b.json profiles/r03_pmc_traffic_decode_qwen3_0_6b.json      # (on the box: so that the 0.6B bench line below reads it)
timeout 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
for mdl in qwen3-0.6b qwen3.5-0.8b qwen3-vl-2b qwen3.8-27b; do
  timeout 600 python bench.py --model $mdl $([ $mdl = qwen3.8-27b ] && echo "--steps 32 --warmup 4") > $OUT/bench_$mdl.json 2> $OUT/bench_$mdl.err
done
timeout 300 python tools/bench_engine.py qwen3-8b 256 128 128 8 32,64,128 > $OUT/eng.log 2>&1; grep "tok/s" $OUT/eng.log | cut -c1-150
kt() { local n=$1; shift
    timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/kt_$n -o $n -- "$@" > $OUT/kt_$n.log 2>&1
    python tools/rocpd_stats.py $(ls $OUT/kt_$n/*_results.db | head -1) $OUT/${n}_kernel_stats.csv > /dev/null 2>>$OUT/kt_$n.log
    rm -rf $OUT/kt_$n; }
kt decode_qwen3_8b_engine python bench.py --no-cpu-baseline --steps 32 --warmup 4
kt decode_qwen3_0p6b_engine python bench.py --model qwen3-0.6b --no-cpu-baseline --steps 64 --warmup 4
kt decode_qwen3_vl_2b python bench.py --model qwen3-vl-2b --no-cpu-baseline --steps 32 --warmup 4
kt decode_qwen3_8_27b python bench.py --model qwen3.8-27b --no-cpu-baseline --steps 16 --warmup 2
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        for l in open(f):
            if l.startswith("{"):
                d = json.loads(l); print(f.split("/")[-1], d["value"], d["ms_per_step"], d["roofline_step"].get("frac"), d["roofline"].get("us_per_launch"), d["roofline"].get("traffic"), (d.get("parity") or {}).get("logit_rel"), (d.get("parity") or {}).get("ok"), (d.get("prefill") or {}).get("ms"), (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "ERR", e)
PY
