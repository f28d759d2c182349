"""Image + text path at its real size (BASELINE configs[3]): a 448 x 448 RGB image through the preprocessor
(cm_image_preprocess -> grid 1 x 28 x 28 = 784 patches), the 24 x 1024 vision tower (+ DeepStack mergers for Qwen3-VL) -> 196
merged tokens, the image+text prefill and a few decode steps.
   python tools/bench_vision.py [reps] [qwen3-vl-2b | qwen3.5-vl-0.8b]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crane_amd import configs
from crane_amd.backend import Model

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
name = sys.argv[2] if len(sys.argv) > 2 else "qwen3-vl-2b"
cfg = configs.get_config(name)
m = Model.synthetic(cfg, seed=0, max_seq_len=2048, max_seqs=2)
grid = [[1, 28, 28]]
from crane_amd.processor import PreprocessorConfig
rng = np.random.default_rng(0)
image = rng.integers(0, 256, size=(448, 448, 3), dtype=np.uint8)
t0 = time.perf_counter()
pix, g = PreprocessorConfig().process(image)
t_pp = time.perf_counter() - t0
assert g == (1, 28, 28) and pix.shape == (784, 1536)
feat = m.encode_images(pix, grid)            # warm-up (allocates the tower scratch)
t0 = time.perf_counter()
for _ in range(reps):
    feat = m.encode_images(pix, grid)
t_enc = (time.perf_counter() - t0) / reps
img = cfg["image_token_id"]
ids = [5, 6, 7, cfg["vision_start_token_id"]] + [img] * 196 + [cfg["vision_end_token_id"], 8, 9, 10]
m.clear_kv_cache()
m.vlm_forward(ids, pix, grid)
m.clear_kv_cache()
t0 = time.perf_counter()
logits, nxt = m.vlm_forward(ids, pix, grid)
t_pre = time.perf_counter() - t0
toks, pos = [nxt], len(ids)
t0 = time.perf_counter()
for _ in range(16):
    toks.append(m.forward_step_greedy([toks[-1]], pos)); pos += 1
t_dec = (time.perf_counter() - t0) / 16
vc = cfg["vision_config"]
flops = 784 * vc["depth"] * (2 * (3 * vc["hidden_size"] ** 2 + vc["hidden_size"] ** 2 + 2 * vc["hidden_size"] * vc["intermediate_size"])) \
        + vc["depth"] * 4 * 784 * 784 * vc["hidden_size"]
print({"model": name, "preprocess_ms": round(t_pp * 1e3, 3), "tower_ms": round(t_enc * 1e3, 3), "tower_tflops": round(flops / t_enc / 1e12, 1), "patches": 784, "merged_tokens": int(feat.shape[0]),
       "vlm_prefill_ms": round(t_pre * 1e3, 3), "prompt_tokens": len(ids), "decode_ms_per_token": round(t_dec * 1e3, 3)})
m.close()
