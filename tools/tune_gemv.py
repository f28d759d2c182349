#!/usr/bin/env python3
"""GPU micro-benchmark of the GEMV variants (CM_GEMV_CFG / CM_GEMV_BLOCKS_PER_CU) on Qwen3-8B shapes."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json
sys.path.insert(0, %r)
from crane_amd import configs
from crane_amd.backend import Model
m = Model.synthetic(configs.get_config(sys.argv[1]), seed=0, max_seq_len=2048, max_seqs=1)
out = {}
for k in ("qkv", "o", "gate_up", "down", "lm_head"):
    r = m.bench_kernel(k, 360 if k != "lm_head" else 60)
    out[k] = (round(r["ms"] * 1e3, 2), round(r["bytes"] / r["ms"] / 1e6, 0))
print(json.dumps(out))
''' % ROOT
model = sys.argv[1] if len(sys.argv) > 1 else "qwen3-8b"
cfgs = [None, "2,8,0", "2,8,1", "2,4,0", "2,4,1", "4,4,0", "4,2,1", "4,2,0", "8,2,0", "8,1,1", "2,2,1"]
for cfg in cfgs:
    for bpc in (0, 2, 3, 4, 6, 8):
        env = dict(os.environ)
        if cfg: env["CM_GEMV_CFG"] = cfg
        elif bpc: continue
        if bpc: env["CM_GEMV_BLOCKS_PER_CU"] = str(bpc)
        try:
            r = subprocess.run([sys.executable, "-c", CHILD, model], env=env, capture_output=True, text=True, timeout=300)
            line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr.strip()[-300:]
        except Exception as e:
            line = str(e)
        print(f"cfg={cfg} bpc={bpc}: {line}", flush=True)
