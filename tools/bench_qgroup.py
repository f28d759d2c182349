"""Decode-group step of an in-situ quantised model (cm_decode_batch over int8-MFMA GEMMs): ms per step at NB sequences, short
context, L layers of Qwen3-8B width -- the projections dominate.  tools/bench_qgroup.py [layers] [nb] [isq]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crane_amd import configs
from crane_amd.backend import Model
L = int(sys.argv[1]) if len(sys.argv) > 1 else 8
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 128
ISQ = sys.argv[3] if len(sys.argv) > 3 else "q8_0"
cfg = dict(configs.get_config("qwen3-8b"), num_hidden_layers=L)
V = cfg["vocab_size"]
kw = {} if os.environ.get("QG_QP") else {"quant_prefill": False}
m = Model.synthetic(cfg, seed=0, max_seq_len=int(os.environ.get("QG_MSL", "128")), isq=ISQ, max_seqs=NB + 1, **kw)
seqs = []
for b in range(NB):
    s = m.seq_alloc(); m.seq_forward(s, [(7 * i + 3 + 11 * b) % V for i in range(8)], 0, want_logits=False); seqs.append(s)
toks = [(5 + 3 * b) % V for b in range(NB)]
for _ in range(3):
    _, g = m.step_batch_decode(seqs, toks, want_logits=False); toks = [int(t) % V for t in g]
K = 20
t0 = time.perf_counter()
for _ in range(K):
    _, g = m.step_batch_decode(seqs, toks, want_logits=False); toks = [int(t) % V for t in g]
dt = (time.perf_counter() - t0) / K
print(f"qgroup L={L} nb={NB} {ISQ}: {dt * 1e3:7.3f} ms/step = {dt * 1e6 / L:7.1f} us/layer (head included)  ids {toks[:4]}", flush=True)
m.close()
