"""Prompt-pass latency of synthetic Qwen3-8B over prompt lengths.   usage: tools/prefill_sweep.py [prefill_split] [lengths]
prefill_split 0 = parity mode (activations as bf16 hi + lo, two MFMAs per product), 1 = plain bf16 activations (one MFMA;
what a bf16 GPU forward of the reference computes).  With prefill_split = 1 the last-token logits are also compared with
the parity mode's on the longest prompt."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from crane_amd import configs
from crane_amd.backend import Model
SPLIT = int(sys.argv[1]) if len(sys.argv) > 1 else 0
LENS = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [32, 64, 128, 200, 256, 384, 512, 1024]
cfg = configs.get_config("qwen3-8b")
m = Model.synthetic(cfg, seed=0, max_seq_len=max(LENS) + 64, max_seqs=2, prefill_split=SPLIT)
for n in LENS:
    ids = configs.synthetic_prompt(n, cfg["vocab_size"])
    m.clear_kv_cache(); m.forward_step_greedy(ids, 0)
    ts = []
    for _ in range(3):
        m.clear_kv_cache(); t0 = time.perf_counter(); m.forward_step_greedy(ids, 0); ts.append(time.perf_counter() - t0)
    print(f"prefill_split={SPLIT} prefill {n:5d} tokens: {min(ts) * 1e3:7.2f} ms  ({n / min(ts):9.0f} tok/s)", flush=True)
if SPLIT:
    ids = configs.synthetic_prompt(max(LENS), cfg["vocab_size"])
    m.clear_kv_cache(); a = np.asarray(m.forward_step(ids, 0)).reshape(-1).astype(np.float64)
    m.close()
    m = Model.synthetic(cfg, seed=0, max_seq_len=max(LENS) + 64, max_seqs=2, prefill_split=0)
    b = np.asarray(m.forward_step(ids, 0)).reshape(-1).astype(np.float64)
    print(f"last-token logits, plain bf16 vs parity mode at {max(LENS)} tokens: max|d| / max|ref| = {np.abs(a - b).max() / np.abs(b).max():.3e}, "
          f"argmax equal: {int(a.argmax()) == int(b.argmax())}")
m.close()
