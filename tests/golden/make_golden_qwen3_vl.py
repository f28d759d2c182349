#!/usr/bin/env python3
"""Generate tests/golden/qwen3_vl_tiny.npz with HF Qwen3VLForConditionalGeneration (CPU, f32, eager): vision-tower output,
the DeepStack feature maps, VLM prefill logits and the greedy continuation for one synthetic 4x6-patch image.
HF's PatchMerger uses erf-GELU; the reference's `xs.gelu()` (qwen3_vl/vision.rs:276) is candle's tanh form, so the fixture
stores the HF (erf) result and the tests run the product with CM_VISION_MERGER_GELU=erf against it, and the default
(reference) mode against the oracle.  Run from the repo root."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from crane_amd import configs, synth  # noqa: E402
from transformers import Qwen3VLConfig, Qwen3VLForConditionalGeneration  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def real_tower_config():
    """Qwen3-VL-2B (BASELINE configs[3]) with the REAL vision tower (depth 24, hidden 1024, 16 heads of 64, 2304 position
    embeddings, DeepStack after blocks 5 / 11 / 17) and the real text widths, the text depth cut to 4 layers (DeepStack feeds
    the first three) and the vocabulary kept: what tests/test_qwen3_vl.py pins the oracle on at full tower size."""
    cfg = configs.get_config("qwen3-vl-2b")
    cfg = dict(cfg, text_config=dict(cfg["text_config"], num_hidden_layers=4, max_position_embeddings=4096))
    return cfg


def main():
    for tag in sys.argv[1:] or ("tiny", "tower24"):
        one(tag)


def one(tag):
    cfg = configs.get_config("tiny-qwen3-vl") if tag == "tiny" else real_tower_config()
    w = synth.synth_weights_f32(cfg, 0)
    tc = {k: v for k, v in cfg["text_config"].items() if k not in ("model_type", "torch_dtype")}
    vc = {k: v for k, v in cfg["vision_config"].items() if k != "model_type"}
    hc = Qwen3VLConfig(text_config=tc, vision_config=vc, image_token_id=cfg["image_token_id"], video_token_id=cfg["video_token_id"],
                       tie_word_embeddings=cfg["tie_word_embeddings"], vision_start_token_id=cfg["vision_start_token_id"],
                       vision_end_token_id=cfg["vision_end_token_id"])
    hc._attn_implementation = hc.text_config._attn_implementation = hc.vision_config._attn_implementation = "eager"
    m = Qwen3VLForConditionalGeneration(hc).float().eval()
    res = m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w.items()}, strict=False)
    assert not res.unexpected_keys and all(k == "lm_head.weight" and cfg["tie_word_embeddings"] for k in res.missing_keys), res
    grid = [[1, 4, 6]] if tag == "tiny" else [[1, 8, 12]]
    n_patch = grid[0][1] * grid[0][2]
    pix = np.random.default_rng(0).standard_normal((n_patch, 3 * 2 * 16 * 16)).astype(np.float32)
    IMG = cfg["image_token_id"]
    ids = [3, 10, cfg["vision_start_token_id"]] + [IMG] * (n_patch // 4) + [cfg["vision_end_token_id"], 17, 24, 31]
    kw = dict(input_ids=torch.tensor([ids]), pixel_values=torch.from_numpy(pix), image_grid_thw=torch.tensor(grid))
    try:                                                   # newer processors pass the modality of every token explicitly
        kw_mm = dict(kw, mm_token_type_ids=torch.tensor([[1 if t == IMG else 0 for t in ids]]))
        with torch.no_grad():
            out = m(**kw_mm)
        kw = kw_mm
    except TypeError:
        with torch.no_grad():
            out = m(**kw)
    with torch.no_grad():
        vis = m.model.visual(torch.from_numpy(pix), grid_thw=torch.tensor(grid))
        if isinstance(vis, tuple):
            feat, deep = vis[0], list(vis[1])
        else:
            feat, deep = vis.pooler_output, list(vis.deepstack_features)
        toks = m.generate(**kw, max_new_tokens=6, do_sample=False)[0].tolist()
    # (tower24: the test regenerates the pixels from the same rng instead of storing them)
    np.savez_compressed(os.path.join(OUT, f"qwen3_vl_{tag}.npz"), pixel_values=pix if tag == "tiny" else np.zeros((0,), np.float32),
                        grid_thw=np.array(grid), input_ids=np.array(ids),
                        features=feat.numpy().astype(np.float32), deepstack=np.stack([d.numpy() for d in deep]).astype(np.float32),
                        prefill_logits=out.logits[0, -1].numpy().astype(np.float32), greedy_tokens=np.array(toks), seed=np.array([0]))
    print(tag, "features", feat.shape, "deepstack", len(deep), "tokens", toks[len(ids):])


if __name__ == "__main__":
    main()
