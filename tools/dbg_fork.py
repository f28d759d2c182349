import sys; sys.path.insert(0, ".")
import numpy as np
from crane_amd import configs
from crane_amd.backend import Model
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for name in ["tiny-qwen3-untied", "tiny-qwen3.5"]:
    cfg = configs.get_config(name)
    m = Model.synthetic(cfg, seed=0, max_seq_len=256, max_seqs=nf + 2, kv_dtype="f32")
    ids = configs.synthetic_prompt(70, cfg["vocab_size"])
    m.forward_step(ids, 0)
    seqs = [0] + [m.seq_fork(0) for _ in range(nf - 1)]
    lg, _ = m.step_batch_decode(seqs, [5] * nf)
    a = lg[0].reshape(-1)
    print(name, nf, [float(np.abs(a - lg[i].reshape(-1)).max() / np.abs(a).max()) for i in range(1, nf)])
    m.close()
