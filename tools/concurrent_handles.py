"""Two independent handles on ONE device driven from two host threads at once (no tensor parallelism): each must reproduce
what it computes alone.  (Reproducer for the split-K prompt GEMM under concurrent kernels of another stream.)"""
import os, sys, threading
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, ".")
import numpy as np
from crane_amd import configs
from crane_amd.backend import Model
cfgs = [configs.get_config("tiny-qwen3-untied"), dict(configs.get_config("tiny-qwen3-untied"), intermediate_size=768, num_attention_heads=4)]
ids = configs.synthetic_prompt(21, 512)
warm = "warm" in sys.argv
ms = [Model.synthetic(c, seed=i, max_seq_len=256, max_seqs=4) for i, c in enumerate(cfgs)]
if warm:
    base = [m.forward_step(ids, 0)[0, 0].copy() for m in ms]
else:
    tmp = [Model.synthetic(c, seed=i, max_seq_len=256, max_seqs=4) for i, c in enumerate(cfgs)]
    base = [m.forward_step(ids, 0)[0, 0].copy() for m in tmp]
    for m in tmp: m.close()
bad = [0, 0]
bar = threading.Barrier(2)
def work(i):
    for it in range(30):
        bar.wait()
        m = ms[i]
        m.clear_kv_cache()
        a = m.forward_step(ids, 0)[0, 0]
        r = float(np.abs(a - base[i]).max() / np.abs(base[i]).max())
        if r > 1e-6: bad[i] += 1; print(f"thread {i} iter {it}: rel {r:.2e}", flush=True)
th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
[t.start() for t in th]; [t.join() for t in th]
print("mismatches:", bad)
