#!/bin/bash
# decode-round launch fusions and the unrolled split-K reductions: correctness subset, then the serving loop, the vision tower and the
# decode-group GEMMs
OUT=$1
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -x -k "wide_gemm or large_decode_groups or engine or batch or vl or vision or prefill" 2>&1 | tail -4
timeout 300 python tools/bench_engine.py qwen3-8b 256 128 128 8 64,128 2>&1 | grep "tok/s" | cut -c1-170
timeout 200 python tools/bench_vision.py 20 qwen3-vl-2b 2>&1 | tail -1
timeout 200 python tools/bench_gemm.py 128 0 3 2>&1 | tail -4
