import sys, json
sys.path.insert(0, '/root/repo')
from crane_amd import configs
from crane_amd.backend import Model
for L in (1, 2, 8):
    cfg = dict(configs.get_config("qwen3.8-27b"), num_hidden_layers=L, vocab_size=4096, full_attention_interval=1 if L < 4 else 4)
    try:
        m = Model.synthetic(cfg, seed=0, max_seq_len=2048, max_seqs=1)
    except Exception as e:
        print(L, "ERR", e); continue
    out = {}
    for k in ("o", "gate_up", "down"):
        r = m.bench_kernel(k, 200)
        out[k] = (round(r["ms"] * 1e3, 2), round(r["bytes"] / r["ms"] / 1e6, 0))
    print("layers", L, out, flush=True)
    m.close()
