#!/bin/bash
# bisect which allocation of a handle is read before it is written (CM_DEBUG_POISON): usage poison_bisect.sh <seq> [byte]
SEQ=${1:-d2}; B=${2:-0x3F}
echo "single under poison:"; CM_DEBUG_POISON=$B timeout 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bad() { CM_DEBUG_POISON=$B,$1,$2 timeout 60 python tools/tp_group_debug.py $SEQ 2>&1 | tail -1 | grep -q "e-0[5-9]\|0.00e+00" && return 1 || return 0; }
lo=0; hi=400
if ! bad $lo $hi; then echo "range 0..400 is clean"; exit 0; fi
while [ $lo -lt $hi ]; do
  mid=$(( (lo + hi) / 2 ))
  if bad $lo $mid; then hi=$mid; else lo=$((mid + 1)); fi
  echo "  -> [$lo, $hi]"
done
echo "first offending allocation index: $lo"
CM_DEBUG_POISON=$B,$lo,$lo timeout 60 python tools/tp_group_debug.py $SEQ 2>&1 | tail -1
