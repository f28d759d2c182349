#!/usr/bin/env python3
"""GEMV variant sweep (CM_GEMV_CFG / CM_GEMV_BLOCKS_PER_CU) on the Qwen3.8-27B shapes (K = 5120 / 6144 / 17408), 8 layers."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json
sys.path.insert(0, %r)
from crane_amd import configs
from crane_amd.backend import Model
cfg = dict(configs.get_config("qwen3.8-27b"), num_hidden_layers=8, vocab_size=4096)
m = Model.synthetic(cfg, seed=0, max_seq_len=2048, max_seqs=1)
out = {}
for k in ("qkv", "o", "gate_up", "down"):
    r = m.bench_kernel(k, 160)
    out[k] = (round(r["ms"] * 1e3, 2), round(r["bytes"] / r["ms"] / 1e6, 0))
print(json.dumps(out))
''' % ROOT
cfgs = sys.argv[1].split(";") if len(sys.argv) > 1 else [None, "4,2,1", "4,2,0", "8,2,0", "8,2,1", "8,1,0"]
for cfg in cfgs:
    for bpc in ([int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else (0, 2)):
        env = dict(os.environ)
        if cfg and cfg != "-": env["CM_GEMV_CFG"] = cfg
        if bpc: env["CM_GEMV_BLOCKS_PER_CU"] = str(bpc)
        try:
            r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
            line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr.strip()[-300:]
        except Exception as e:
            line = str(e)
        print(f"cfg={cfg} bpc={bpc}: {line}", flush=True)
