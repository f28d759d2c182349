#!/bin/bash
# A/B of two builds of the library on one box (alternating): crane_amd/lib_base.bin vs crane_amd/lib_new.bin
#   bash tools/probes/ab_lib.sh <outdir> -- <cmd ...>     (cmd is run with each library in place; '--' separated list of commands by ';;')
OUT=$1; shift; shift
cp crane_amd/libcrane_mi355.so /tmp/lib_orig.so
for v in base new base new; do
  cp crane_amd/lib_$v.bin crane_amd/libcrane_mi355.so
  echo "--- $v"
  eval "$@" 2>&1 | grep -v "^$" | tail -12 | cut -c1-220
done
cp /tmp/lib_orig.so crane_amd/libcrane_mi355.so
