#!/bin/bash
# engine timeline (tools/engine_trace.py) with the base and the new library
OUT=$1
for v in base new; do
  cp crane_amd/lib_$v.bin crane_amd/libcrane_mi355.so
  echo "=========== $v"
  timeout 300 python tools/engine_trace.py qwen3-8b 2>&1 | sed -n '/traced launch 1/,$p'
done
echo "=========== new GBLK=2"
CM_ENG_GBLK=2,0,0,0 timeout 300 python tools/engine_trace.py qwen3-8b 2>&1 | sed -n '/traced launch 1/,$p'
cp crane_amd/lib_new.bin crane_amd/libcrane_mi355.so
