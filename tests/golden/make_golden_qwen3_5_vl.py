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


def real_config():
    """The reference's LIVE image path at real size: the 24 x 1024 tower of `qwen3.5-vl-0.8b` in front of the Qwen3.5-0.8B text
    model (H 1024, 8 q / 2 kv heads of 256, 16 GDN heads of 128, I 3584, 248 320-entry tied table) cut to 4 layers (3 GDN + 1
    gated attention)."""
    cfg = configs.get_config("qwen3.5-vl-0.8b")
    return dict(cfg, text_config=dict(cfg["text_config"], num_hidden_layers=4, max_position_embeddings=4096))


def main():
    for tag in sys.argv[1:] or ("tiny", "tower24"):
        one(tag)


def one(tag):
    cfg = configs.get_config("tiny-qwen3.5-vl") if tag == "tiny" else real_config()
    w = synth.synth_weights_f32(cfg, 0)
    tc = {k: v for k, v in cfg["text_config"].items() if k not in ("model_type", "torch_dtype", "full_attention_interval", "attn_output_gate")}
    vc = {k: v for k, v in cfg["vision_config"].items() if k != "model_type"}
    tied = bool(cfg.get("tie_word_embeddings", False))
    hc = Qwen3_5Config(text_config=tc, vision_config=vc, image_token_id=cfg["image_token_id"], tie_word_embeddings=tied,
                       vision_start_token_id=cfg["vision_start_token_id"], vision_end_token_id=cfg["vision_end_token_id"])
    hc._attn_implementation = hc.text_config._attn_implementation = hc.vision_config._attn_implementation = "eager"
    m = Qwen3_5ForConditionalGeneration(hc).float().eval()
    res = m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w.items()}, strict=False)
    assert not res.unexpected_keys and all(k == "lm_head.weight" and tied for k in res.missing_keys), res
    grid = [[1, 4, 6]] if tag == "tiny" else [[1, 8, 12]]
    n_patch = grid[0][1] * grid[0][2]
    pix = np.random.default_rng(0).standard_normal((n_patch, 3 * 2 * 16 * 16)).astype(np.float32)
    IMG = cfg["image_token_id"]
    ids = [3, 10, IMG - 1] + [IMG] * (n_patch // 4) + [IMG + 1, 17, 24, 31]
    mm = torch.tensor([[1 if t == IMG else 0 for t in ids]])
    with torch.no_grad():
        feat = m.model.visual(torch.from_numpy(pix), grid_thw=torch.tensor(grid))
        feat = feat[0] if isinstance(feat, tuple) else getattr(feat, "pooler_output", feat)
        out = m(input_ids=torch.tensor([ids]), pixel_values=torch.from_numpy(pix), image_grid_thw=torch.tensor(grid), mm_token_type_ids=mm)
        toks = m.generate(input_ids=torch.tensor([ids]), pixel_values=torch.from_numpy(pix), image_grid_thw=torch.tensor(grid),
                          mm_token_type_ids=mm, max_new_tokens=6, do_sample=False)[0].tolist()
    # (tower24: the test regenerates the pixels from the same rng instead of storing them)
    np.savez_compressed(os.path.join(OUT, f"qwen3_5_vl_{tag}.npz"), pixel_values=pix if tag == "tiny" else np.zeros((0,), np.float32),
                        grid_thw=np.array(grid), input_ids=np.array(ids),
                        features=feat.numpy().astype(np.float32), prefill_logits=out.logits[0, -1].numpy().astype(np.float32),
                        greedy_tokens=np.array(toks), seed=np.array([0]))
    print(tag, "features", feat.shape, "tokens", toks[len(ids):])


if __name__ == "__main__":
    main()
