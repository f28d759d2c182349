"""Prompt-pass GEMMs of one Qwen3-8B layer, old (128-row, register-staged) vs new (256-row, LDS-DMA) kernel, interleaved in ONE process
(within-probe A/B).   usage: tools/bench_gemm.py [rows=1024] [split=0|1] [rounds=3]
split 0 = parity mode (bf16 hi + lo activations), 1 = plain bf16.  Prints us per launch and useful TFLOP/s per projection."""
import sys
sys.path.insert(0, ".")
from crane_amd import configs
from crane_amd.backend import Model

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
SPLIT = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ROUNDS = int(sys.argv[3]) if len(sys.argv) > 3 else 3
import os
cfg = dict(configs.get_config("qwen3-8b"), num_hidden_layers=int(os.environ.get("LAYERS", "8")))      # LAYERS=1: the weights of a projection stay in the 256 MB Infinity Cache between launches
m = Model.synthetic(cfg, seed=0, max_seq_len=max(M, 2048) + 64, max_seqs=1, prefill_split=SPLIT, prefill_chunk=max(M, 2048))
MODES = (0, 3, 2)                 # cm_debug_set("gemm256"): 0 = 128-row register-staged kernel, 3 = kernels_gemm256.hip, 2 = kernels_gemmw4.hip (round 6)
NAMES = {0: "g128", 3: "g256", 2: "w4"}
res = {}
for r in range(ROUNDS):
    for proj in ("gate_up", "down", "qkv", "o"):
        for mode in MODES:
            m.debug_set("gemm256", mode)
            k = m.bench_kernel(f"pgemm_{proj}@{M}", 40)
            res.setdefault((proj, mode), []).append((k["ms"] * 1e3, k["bytes"]))
for proj in ("gate_up", "down", "qkv", "o"):
    line = f"M={M} split={SPLIT} {proj:8s}"
    for mode in MODES:
        us = sorted(x[0] for x in res[(proj, mode)])
        fl = res[(proj, mode)][0][1]
        line += f" | {NAMES[mode]} {us[len(us) // 2]:8.1f} us (min {us[0]:7.1f}) {fl / (us[len(us) // 2] * 1e-6) / 1e12:7.1f} TF"
    print(line, flush=True)
m.close()
