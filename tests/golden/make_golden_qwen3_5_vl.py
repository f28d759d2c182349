#!/usr/bin/env python3
"""Generate tests/golden/qwen3_5_vl_tiny.npz with HF Qwen3_5ForConditionalGeneration (CPU, f32, eager):
vision-tower output, VLM prefill logits and greedy continuation for one synthetic 4x6-patch image.
HF's PatchMerger uses erf-GELU; the reference's `xs.gelu()` (vision.rs:276) is candle's tanh form, so the
fixture stores the HF (erf) result and the tests run the product with CM_VISION_MERGER_GELU=erf against it,
and the default (reference) mode against the oracle.  Run from the repo root."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from crane_amd import configs, synth  # noqa: E402
from transformers import Qwen3_5Config, Qwen3_5ForConditionalGeneration  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    cfg = configs.get_config("tiny-qwen3.5-vl")
    w = synth.synth_weights_f32(cfg, 0)
    tc = {k: v for k, v in cfg["text_config"].items() if k not in ("model_type", "torch_dtype", "full_attention_interval", "attn_output_gate")}
    vc = {k: v for k, v in cfg["vision_config"].items() if k != "model_type"}
    hc = Qwen3_5Config(text_config=tc, vision_config=vc, image_token_id=cfg["image_token_id"], tie_word_embeddings=False,
                       vision_start_token_id=cfg["vision_start_token_id"], vision_end_token_id=cfg["vision_end_token_id"])
    hc._attn_implementation = hc.text_config._attn_implementation = hc.vision_config._attn_implementation = "eager"
    m = Qwen3_5ForConditionalGeneration(hc).float().eval()
    res = m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w.items()}, strict=False)
    assert not res.missing_keys and not res.unexpected_keys, res
    grid = [[1, 4, 6]]
    pix = np.random.default_rng(0).standard_normal((24, 3 * 2 * 16 * 16)).astype(np.float32)
    IMG = cfg["image_token_id"]
    ids = [3, 10, IMG - 1] + [IMG] * 6 + [IMG + 1, 17, 24, 31]
    mm = torch.tensor([[1 if t == IMG else 0 for t in ids]])
    with torch.no_grad():
        feat = m.model.visual(torch.from_numpy(pix), grid_thw=torch.tensor(grid))
        feat = feat[0] if isinstance(feat, tuple) else getattr(feat, "pooler_output", feat)
        out = m(input_ids=torch.tensor([ids]), pixel_values=torch.from_numpy(pix), image_grid_thw=torch.tensor(grid), mm_token_type_ids=mm)
        toks = m.generate(input_ids=torch.tensor([ids]), pixel_values=torch.from_numpy(pix), image_grid_thw=torch.tensor(grid),
                          mm_token_type_ids=mm, max_new_tokens=6, do_sample=False)[0].tolist()
    np.savez_compressed(os.path.join(OUT, "qwen3_5_vl_tiny.npz"), pixel_values=pix, grid_thw=np.array(grid), input_ids=np.array(ids),
                        features=feat.numpy().astype(np.float32), prefill_logits=out.logits[0, -1].numpy().astype(np.float32),
                        greedy_tokens=np.array(toks), seed=np.array([0]))
    print("features", feat.shape, "tokens", toks[len(ids):])


if __name__ == "__main__":
    main()
