#!/bin/bash
timeout 300 python tools/engine_trace.py qwen3-8b 2>&1 | sed -n '/traced launch 1/,$p'
timeout 300 python tools/engine_trace.py qwen3-0.6b 2>&1 | sed -n '/traced launch 1/,$p' | head -40
