"""Vision tower with several images per call (cm_vision_encode, n_images > 1): ms per call and per image, 448 x 448 images
(784 patches each; <= 4096 patches per call).   python tools/bench_vit_batch.py [qwen3-vl-2b] [1,2,4,5]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crane_amd import configs
from crane_amd.backend import Model
from crane_amd.processor import PreprocessorConfig
name = sys.argv[1] if len(sys.argv) > 1 else "qwen3-vl-2b"
counts = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4, 5]
cfg = configs.get_config(name)
cfg = dict(cfg, text_config=dict(cfg["text_config"], num_hidden_layers=2))          # the tower is what is timed
m = Model.synthetic(cfg, seed=0, max_seq_len=256, max_seqs=1)
rng = np.random.default_rng(0)
pix1, g = PreprocessorConfig().process(rng.integers(0, 256, size=(448, 448, 3), dtype=np.uint8))
one = m.encode_images(pix1, [list(g)])
for n in counts:
    pix = np.concatenate([pix1] * n, axis=0)
    grid = [list(g)] * n
    f = m.encode_images(pix, grid)
    assert np.array_equal(f[: one.shape[0]], one) or np.abs(f[: one.shape[0]] - one).max() / np.abs(one).max() < 1e-4
    t0 = time.perf_counter()
    for _ in range(5):
        m.encode_images(pix, grid)
    dt = (time.perf_counter() - t0) / 5
    print(f"{name}: {n} image(s) per call, {pix.shape[0]} patches: {dt * 1e3:.3f} ms per call = {dt * 1e3 / n:.3f} ms per image", flush=True)
m.close()
