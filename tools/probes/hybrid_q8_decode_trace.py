"""Single-sequence decode of Qwen3.5-2B ISQ Q8_0 (the reference's published configuration) for rocprofv3 --kernel-trace."""
import sys
sys.path.insert(0, ".")
from crane_amd import configs
from crane_amd.backend import Model
cfg = configs.get_config(sys.argv[1] if len(sys.argv) > 1 else "qwen3.5-2b")
m = Model.synthetic(cfg, seed=0, max_seq_len=4352, max_seqs=2, isq="q8_0")
m.debug_fill_kv(2048, seed=1)
toks, _ = m.bench_decode(3, 16)
toks, ms = m.bench_decode(int(toks[-1]), 64)
print("ms/token", ms / 64)
m.close()
