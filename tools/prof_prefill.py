"""One model, a few 1024-token prefills (MFMA GEMM + flash attention): the workload of the MFMA-busy PMC pass."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crane_amd import configs
from crane_amd.backend import Model

name = sys.argv[1] if len(sys.argv) > 1 else "qwen3-8b"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
cfg = configs.get_config(name)
m = Model.synthetic(cfg, seed=0, max_seq_len=4096, max_seqs=1)
ids = configs.synthetic_prompt(n, cfg["vocab_size"])
for _ in range(3):
    m.clear_kv_cache()
    m.forward_step_greedy(ids, 0)
m.close()
