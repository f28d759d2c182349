"""Debug: int8-MFMA decode-group path vs the batched integer-dot GEMV, per row, on a 1- or 2-layer 8B-width ISQ model."""
import sys
sys.path.insert(0, ".")
import numpy as np
from crane_amd import configs
from crane_amd.backend import Model
L = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cfg = dict(configs.get_config("qwen3-8b-2l"), num_hidden_layers=L)
V = cfg["vocab_size"]
m = Model.synthetic(cfg, seed=0, max_seq_len=64, isq="q8_0", max_seqs=3 * nb + 2, quant_prefill=False)
seqs, tw, tw2 = [], [], []
for b in range(nb):
    s = m.seq_alloc()
    m.seq_forward(s, [(7 * i + 3 + 11 * b) % V for i in range(3 + b % 7)], 0, want_logits=False)
    seqs.append(s); tw.append(m.seq_fork(s)); tw2.append(m.seq_fork(s))
toks = [(5 + 3 * b) % V for b in range(nb)]
def rel(a, r): return float(np.abs(a - r).max() / np.abs(r).max())
m.debug_set("q_gemm_min", 0)
want, wgr = m.step_batch_decode(tw, toks)
print("gemv path greedy != argmax rows:", [b for b in range(nb) if int(wgr[b]) != int(want[b, 0].argmax())][:20])
m.debug_set("q_gemm_min", 33)
got, ggr = m.step_batch_decode(seqs, toks)
print("gemm path greedy != argmax rows:", [b for b in range(nb) if int(ggr[b]) != int(got[b, 0].argmax())][:20], [int(x) for x in ggr[:8]])
m.debug_set("q_gemm_min", 0)
f32, _ = m.step_batch_decode(tw2, toks)
r1 = [rel(got[b, 0], want[b, 0]) for b in range(nb)]
r2 = [rel(want[b, 0], f32[b, 0]) for b in range(nb)]
r3 = [rel(got[b, 0], f32[b, 0]) for b in range(nb)]
print("layers", L, "nb", nb)
print("gemm vs gemv  : max %.3e median %.3e" % (max(r1), float(np.median(r1))), ["%.1e" % x for x in r1[:12]])
print("gemv vs gemv again: max %.3e median %.3e" % (max(r2), float(np.median(r2))))
print("gemm vs gemv again: max %.3e median %.3e" % (max(r3), float(np.median(r3))))
m.close()
