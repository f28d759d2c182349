import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crane_amd import configs, synth
from crane_amd.backend import Model
from oracle import qwen3_5_oracle as O
base = configs.get_config("tiny-qwen3.5")
def rel(a, r): return float(np.abs(a - r).max() / np.abs(r).max())
for L in (1, 3, 4, 8):
    cfg = dict(base, num_hidden_layers=L)
    w = synth.synth_weights_f32(cfg, 0)
    o = O.Qwen35Oracle(O.Qwen35Config.from_json(cfg), w)
    m = Model.synthetic(cfg, seed=0, max_seq_len=128, max_seqs=2, kv_dtype="f32")
    errs = []
    for pos, t in enumerate([3, 10, 17, 24, 31]):
        errs.append(rel(m.forward_step([t], pos)[0, 0], o.forward([t], pos)))
    print("L", L, ["%.2e" % e for e in errs], flush=True)
    m.close()
