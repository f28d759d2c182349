"""GPU-box reproducer for order-dependent failures of tests/test_gpu_tp_group.py: groups created and destroyed in a given order.
   python tools/tp_group_debug.py d2,d4,h2,d2   (d = dense tiny, h = hybrid tiny, digit = tp)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, ".")
import numpy as np
from crane_amd import configs
from crane_amd.backend import Model, GenerationConfig
def rel(a, b): return float(np.abs(a - b).max() / np.abs(b).max())
def grp(cfg, tp, **kw):
    kw.setdefault("max_seq_len", 256); kw.setdefault("max_seqs", 4)
    return Model.synthetic(cfg, seed=0, tp_size=tp, tp_in_process=True, tp_devices=[0] * tp, **kw)
def one(cfg, **kw):
    kw.setdefault("max_seq_len", 256); kw.setdefault("max_seqs", 4)
    return Model.synthetic(cfg, seed=0, **kw)
cfgs = {"d": configs.get_config("tiny-qwen3-untied"), "h": configs.get_config("tiny-qwen3.5")}
ids = configs.synthetic_prompt(21, 512)
ref = {}
for name, cfg in cfgs.items():
    s = one(cfg); ref[name] = (s.forward_step(ids, 0)[0, 0], s.forward_step([5], 21)[0, 0], s.generate(ids, GenerationConfig.greedy(16))); s.close()
for item in sys.argv[1].split(","):
    name, tp = item[0], int(item[1])
    t0 = time.time()
    g = grp(cfgs[name], tp, use_graph=-1 if "nograph" in sys.argv else 0)
    try:
        ops = os.environ.get("OPS", "pdg")
        a = g.forward_step(ids, 0)[0, 0]
        b = g.forward_step([5], 21)[0, 0] if "d" in ops else ref[name][1]
        t = g.generate(ids, GenerationConfig.greedy(16)) if "g" in ops else ref[name][2]
        print(f"{item}: prefill {rel(a, ref[name][0]):.2e} decode {rel(b, ref[name][1]):.2e} tokens {'equal' if t == ref[name][2] else 'DIFFER'}  ({time.time() - t0:.2f}s)", flush=True)
    except Exception as e:
        print(f"{item}: ERROR {e} ({time.time() - t0:.2f}s)", flush=True)
    g.close()
