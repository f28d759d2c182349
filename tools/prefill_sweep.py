import sys, time
sys.path.insert(0, ".")
from crane_amd import configs
from crane_amd.backend import Model
cfg = configs.get_config("qwen3-8b")
m = Model.synthetic(cfg, seed=0, max_seq_len=2048, max_seqs=2)
for n in (32, 64, 128, 200, 256, 384, 512, 1024):
    ids = configs.synthetic_prompt(n, cfg["vocab_size"])
    m.clear_kv_cache(); m.forward_step_greedy(ids, 0)
    ts = []
    for _ in range(3):
        m.clear_kv_cache(); t0 = time.perf_counter(); m.forward_step_greedy(ids, 0); ts.append(time.perf_counter() - t0)
    print(f"prefill {n:5d} tokens: {min(ts) * 1e3:7.2f} ms  ({n / min(ts):9.0f} tok/s)", flush=True)
m.close()
