"""Latency of the device sampler at a real vocabulary (qwen3-0.6b: V = 151 936) -- sampling.rs:25-27 quotes ~24 ms
for the host sort it replaces."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from crane_amd import configs
from crane_amd.backend import Model

m = Model.synthetic(configs.get_config("qwen3-0.6b"), seed=0, max_seq_len=512)
ids = configs.synthetic_prompt(16, m.vocab_size)
m.forward_step_greedy(ids, 0)
for name, kw in [("greedy_top1", dict(temperature=0.0)), ("topk40_T0.8", dict(temperature=0.8, top_k=40)),
                 ("topp0.9_T0.8", dict(temperature=0.8, top_p=0.9)), ("full_gumbel", dict(temperature=1.0)),
                 ("topk40+pen64", dict(temperature=0.8, top_k=40, repetition_penalty=1.1, frequency_penalty=0.1))]:
    ctx = list(range(100, 164)) if "pen" in name else []
    for _ in range(5):
        m.sample(ctx, **kw)
    t0 = time.perf_counter()
    n = 200
    for d in range(n):
        m.sample(ctx, draw=d, **kw)
    print(f"{name:16s} {(time.perf_counter() - t0) / n * 1e6:8.1f} us/sample (incl. launch + sync + ctypes)")
