#!/bin/bash
TRACE_XCD=1 timeout 300 python tools/engine_trace.py qwen3-0.6b 2>&1 | sed -n '/traced launch 1/,$p' | grep -v "done by XCD" | head -60
TRACE_XCD=1 timeout 300 python tools/engine_trace.py qwen3-8b 2>&1 | sed -n '/traced launch 1/,$p' | grep -A5 "phase 5 o_proj" | head -40
