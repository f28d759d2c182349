"""Prompt pass over ISQ / Q8_0-layout weights: ms per pass at several prompt lengths (int8 matrix cores by default;
argv[3] = 0: the dequantised-to-bf16 GEMMs, cm_debug_set("prefill_q8", 0))."""
import sys, time
sys.path.insert(0, ".")
from crane_amd import configs
from crane_amd.backend import Model

model = sys.argv[1] if len(sys.argv) > 1 else "qwen3-8b"
isq = sys.argv[2] if len(sys.argv) > 2 else "q8_0"
q8 = int(sys.argv[3]) if len(sys.argv) > 3 else 1
lens = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [128, 512, 1024, 2048]
split = int(sys.argv[5]) if len(sys.argv) > 5 else 0
cfg = configs.get_config(model)
m = Model.synthetic(cfg, seed=0, max_seq_len=max(lens) + 64, max_seqs=2, isq=isq, prefill_split=split)
if not q8:
    m.debug_set("prefill_q8", 0)
V = cfg["vocab_size"]
for n in lens:
    ids = [(7 * i + 3) % V for i in range(n)]
    m.clear_kv_cache(); m.forward_step_greedy(ids, 0)
    ts = []
    for _ in range(3):
        m.clear_kv_cache()
        t0 = time.perf_counter(); m.forward_step_greedy(ids, 0); ts.append(time.perf_counter() - t0)
    print(f"{model} isq {isq} q8 {q8} split {split}: {n} tokens {min(ts) * 1e3:.2f} ms = {n / min(ts):.0f} tok/s", flush=True)
m.close()
