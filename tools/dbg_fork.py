import sys; sys.path.insert(0, ".")
import numpy as np
from crane_amd import configs
from crane_amd.backend import Model
for name in ["tiny-qwen3-untied", "tiny-qwen3.5"]:
    for kv in ["f32", "bf16", "int8"]:
        cfg = configs.get_config(name)
        m = Model.synthetic(cfg, seed=0, max_seq_len=256, max_seqs=3, kv_dtype=kv)
        ids = configs.synthetic_prompt(70, cfg["vocab_size"])
        m.forward_step(ids, 0)
        s1 = m.seq_fork(0)
        lg, _ = m.step_batch_decode([0, s1], [5, 5])
        a, b = lg[0].reshape(-1), lg[1].reshape(-1)
        print(name, kv, float(np.abs(a - b).max() / np.abs(a).max()))
        m.close()
