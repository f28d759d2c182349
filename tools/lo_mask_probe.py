"""Which projections of the prompt pass need the lo plane of their activations?  1024-token prompt of Qwen3-8B (36 layers), last
position's logits with one or more GEMMs on plain bf16 activations against the all-hi+lo pass; and the time of the pass."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crane_amd import configs
from crane_amd.backend import Model
name = sys.argv[1] if len(sys.argv) > 1 else "qwen3-8b"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
cfg = configs.get_config(name)
m = Model.synthetic(cfg, seed=0, max_seq_len=2048, max_seqs=1)
ids = configs.synthetic_prompt(n, cfg["vocab_size"])
def run(mask):
    m.debug_set("prefill_lo_mask", mask)
    m.clear_kv_cache(); lg = m.forward_step(ids, 0)[0, 0].copy()
    m.clear_kv_cache(); t0 = time.perf_counter(); m.forward_step_greedy(ids, 0); dt = time.perf_counter() - t0
    return lg, dt
ref, t_ref = run(0)
print(f"mask 0 (all hi+lo): {t_ref*1e3:.2f} ms")
for mask in (1, 2, 4, 8, 3, 10, 12, 14, 15):
    lg, dt = run(mask)
    names = "+".join(nm for b, nm in ((1, "qkv"), (2, "o"), (4, "gate_up"), (8, "down")) if mask & b)
    print(f"mask {mask:2d} plain: {names:22s} {dt*1e3:6.2f} ms  rel {np.abs(lg-ref).max()/np.abs(ref).max():.3e}  argmax_equal {int(lg.argmax())==int(ref.argmax())}", flush=True)
m.close()
