"""Per-projection GEMV timing on Qwen3-8B shapes (cm_bench_kernel: HIP events, cycling layers so nothing is cache-hot)."""
import sys
sys.path.insert(0, ".")
from crane_amd import configs
from crane_amd.backend import Model

m = Model.synthetic(configs.get_config(sys.argv[1] if len(sys.argv) > 1 else "qwen3-8b"), seed=0, max_seq_len=2048,
                    isq=sys.argv[2] if len(sys.argv) > 2 else None)
for which in ["qkv", "o", "gate_up", "down", "lm_head"]:
    r = m.bench_kernel(which, 360 if which != "lm_head" else 20)
    ms, nbytes = r["ms"], r["bytes"]
    print(f"{which:8s} {ms * 1e3:8.2f} us  {nbytes / ms / 1e9:8.2f} TB/s")
