#!/bin/bash
for i in 1 2; do TRACE_XCD=1 timeout 300 python tools/engine_trace.py qwen3-8b 2>&1 | sed -n '/traced launch 1/,$p' | grep "done by XCD"; done
rocm-smi --showtopo 2>/dev/null | head -5
