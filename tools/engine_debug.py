"""Phase-by-phase comparison of the persistent decode kernel with the launch path on a 1-layer model (debug)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crane_amd import configs
from crane_amd.backend import Model


def rel(a, b):
    return float(np.abs(a - b).max() / max(1e-30, np.abs(b).max()))


name = sys.argv[1] if len(sys.argv) > 1 else "eng-qwen3"
for L in (1, 2):
    cfg = configs.get_config(name)
    cfg["num_hidden_layers"] = L
    H, I = cfg["hidden_size"], cfg["intermediate_size"]
    D, Hq, Hkv = cfg["head_dim"], cfg["num_attention_heads"], cfg["num_key_value_heads"]
    qkv_rows = (Hq + 2 * Hkv) * D
    eng = Model.synthetic(cfg, seed=0, max_seq_len=2048, max_seqs=2, engine=1, use_graph=-1)
    ref = Model.synthetic(cfg, seed=0, max_seq_len=2048, max_seqs=2, engine=-1, use_graph=-1)
    for ctx in (0, 5, 40, 700):
        if ctx:
            eng.debug_fill_kv(ctx, seed=2); ref.debug_fill_kv(ctx, seed=2)
        else:
            eng.clear_kv_cache(); ref.clear_kv_cache()
        a = eng.forward_step([7], ctx)[0, 0]
        b = ref.forward_step([7], ctx)[0, 0]
        out = [f"L={L} ctx={ctx} logits {rel(a, b):.2e}", f"hidden {rel(eng.debug_read('hidden', H), ref.debug_read('hidden', H)):.2e}"]
        if L == 1:
            out.append(f"qkv {rel(eng.debug_read('eng_qkv', qkv_rows), ref.debug_read('qkv', qkv_rows)):.2e}")
            ea, ra = eng.debug_read('eng_attn', Hq * D), ref.debug_read('attn', Hq * D)
            out.append(f"attn {rel(ea, ra):.2e}")
            per_head = np.abs(ea - ra).reshape(Hq, D).max(axis=1) / np.abs(ra).max()
            out.append("worst heads " + str(np.argsort(-per_head)[:4].tolist()) + " " + str(np.round(np.sort(per_head)[-4:], 5).tolist()))
            out.append(f"h {rel(eng.debug_read('eng_h', I), ref.debug_read('hbuf', I)):.2e}")
        print(" | ".join(out), flush=True)
    eng.close(); ref.close()
