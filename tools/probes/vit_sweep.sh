#!/bin/bash
python tools/probes/vit_sweep.py
for c in 128 256 384 512 1024; do CM_KSPLIT_CAP=$c python tools/probes/vit_sweep.py; done
CM_GEMM256=0 python tools/probes/vit_sweep.py
CM_GEMM256=0 CM_KSPLIT_CAP=256 python tools/probes/vit_sweep.py
CM_GEMM256_MIN_BLOCKS=64 python tools/probes/vit_sweep.py
CM_VIT_KSPLIT=2 python tools/probes/vit_sweep.py
