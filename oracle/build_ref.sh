#!/bin/bash
# TEST INFRASTRUCTURE ONLY.  Compiles the two kernel sources of the reference that build from their own file -- the Gated Delta
# Net recurrence (crane-core/kernels/cuda/gdn.cu) and the exact top-k (crane-core/kernels/cuda/topk.cu) -- for gfx950, from
# where they lie under /root/reference, into oracle/_ref/*.hsaco (git-ignored; travels to the GPU box with the snapshot).
# This is how the reference itself builds them on AMD GPUs: ops/rocm.rs:100-124 hands the same source text to hipcc at run
# time with the HIP runtime header force-included ("force-included by candle's shim", gdn.cu:36-40).  The third kernel file,
# fused_ops.cu, needs candle's <cuda_bf16.h> / __shfl_*_sync shim (fused_ops.cu:4-9), which is not under /root/reference:
# treated as unbuildable.  Nothing of the reference is copied into this repository; only code objects are written.
# oracle/ref_kernels.py loads the code objects (hipModuleLoad) for tests/test_gpu_ref_kernels.py.
set -e
REF=${CRANE_REFERENCE:-/root/reference}/crane-core/kernels/cuda
OUT=$(cd "$(dirname "$0")" && pwd)/_ref
[ -d "$REF" ] || { echo "build_ref.sh: $REF not present (GPU box / fresh clone): keeping whatever is in $OUT"; exit 0; }
mkdir -p "$OUT"
for f in gdn topk; do
    if [ ! -f "$OUT/$f.hsaco" ] || [ "$REF/$f.cu" -nt "$OUT/$f.hsaco" ]; then
        hipcc -x hip --offload-arch=gfx950 -O3 -include hip/hip_runtime.h -Wno-pass-failed --genco -o "$OUT/$f.hsaco" "$REF/$f.cu"
    fi
done
ls -la "$OUT"
