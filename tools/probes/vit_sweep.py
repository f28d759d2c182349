"""Vision tower device time (cm_debug_read "vision_ms") at 784 patches.  python tools/probes/vit_sweep.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from crane_amd import configs
from crane_amd.backend import Model
from crane_amd.processor import PreprocessorConfig
cfg = configs.get_config("qwen3-vl-2b")
cfg = dict(cfg, text_config=dict(cfg["text_config"], num_hidden_layers=2))
m = Model.synthetic(cfg, seed=0, max_seq_len=256, max_seqs=1)
rng = np.random.default_rng(0)
pix, g = PreprocessorConfig().process(rng.integers(0, 256, size=(448, 448, 3), dtype=np.uint8))
ts = []
for i in range(8):
    m.encode_images(pix, [list(g)])
    ts.append(float(m.debug_read("vision_ms", 1)[0]))
print(f"tower device ms: min {min(ts[2:]):.3f} med {sorted(ts[2:])[3]:.3f}   env " + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("CM_")), flush=True)
m.close()
