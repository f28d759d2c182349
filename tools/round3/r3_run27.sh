#!/bin/bash
# PMC traffic (separate FETCH_SIZE / WRITE_SIZE passes, kernel-trace only) of the decode step of the other BASELINE configs
OUT=gpurun_out/${1:-r3J}
mkdir -p $OUT
git_rev=$(cat .git_rev 2>/dev/null || echo unknown)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
pmc() { local n=$1 c=$2; shift 2
    timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$n -o $n -- "$@" > $OUT/pmc_$n.log 2>&1
    python tools/pmc_summary.py $OUT/pmc_$n cm:: > $OUT/pmc_$n.json 2>>$OUT/pmc_$n.log
    rm -rf $OUT/pmc_$n; }
for mdl in qwen3-0.6b qwen3.5-0.8b qwen3-vl-2b qwen3.8-27b; do
  san=$(echo $mdl | tr '.-' '__')
  pmc fetch_$san FETCH_SIZE python bench.py --model $mdl --no-cpu-baseline --steps 8 --warmup 2 --no-graph
  pmc write_$san WRITE_SIZE python bench.py --model $mdl --no-cpu-baseline --steps 8 --warmup 2 --no-graph
  python tools/merge_traffic.py $OUT/pmc_fetch_$san.json $OUT/pmc_write_$san.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python bench.py --model $mdl --no-cpu-baseline --steps 8 --warmup 2 --no-graph (tools/r3_run27.sh, source tree $git_rev)" > $OUT/pmc_traffic_decode_$san.json 2>/dev/null
  head -c 900 $OUT/pmc_traffic_decode_$san.json | tail -c 600; echo
done
