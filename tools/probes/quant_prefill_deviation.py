import sys, tempfile, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from crane_amd import configs, synth
from crane_amd.backend import Model
from oracle import gguf_oracle as G
from oracle.qwen3_oracle import Qwen3Oracle, Qwen3Config
import test_gpu_quant as T
def rel(a, r): return float(np.abs(a - r).max() / np.abs(r).max())
cfg = configs.get_config("tiny-qwen3-untied")
w = synth.synth_weights_f32(cfg, seed=0)
d = tempfile.mkdtemp()
for kind in ("q8_0", "q4_k", "mixed"):
    path = os.path.join(d, kind + ".gguf")
    deq = G.write_qwen3_gguf(path, cfg, w, T._types(kind))
    oracle = Qwen3Oracle(Qwen3Config.from_json(cfg), deq)
    ids = configs.synthetic_prompt(70, cfg["vocab_size"])
    ref = oracle.forward(ids, 0)
    out = []
    for sp in (0, 2):
        m = Model.from_pretrained(path, max_seq_len=128, kv_dtype="f32", quant_act="f32", prefill_split=sp)
        got = m.forward_step(ids, 0).reshape(-1); m.close()
        out.append(rel(got, ref))
    print(kind, "prompt logits vs f32 oracle on the dequantised weights: plain bf16 activations %.2e, hi + lo %.2e" % tuple(out))
