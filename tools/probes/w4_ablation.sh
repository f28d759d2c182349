python tools/bench_gemm.py 1024 0 3 2>&1 | grep -E "gate_up|down|qkv| o " 
python tools/bench_gemm.py 1024 1 3 2>&1 | grep -E "gate_up|down|qkv| o "
python tools/bench_gemm.py 128 0 3 2>&1 | grep -E "gate_up|down|qkv| o "
