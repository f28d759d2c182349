"""CPU restatement of the Qwen3-VL image+text path (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).

Reference (dead code there -- `models/mod.rs` does not export it -- but a named BASELINE config):
  vision tower + DeepStack mergers   crane-core/src/models/qwen3_vl/vision.rs:222-278,306-380,558-585 (identical to qwen3_5/vision.rs)
  image rows spliced over the pads   qwen3_vl/mod.rs:86-131
  DeepStack injection                qwen3_vl/text.rs:262-333: after decoder layer i (i < #deepstack) the i-th merger's rows are
                                     ADDED to the hidden states of the visual positions (scatter_add by rank)
  text model                         qwen3_vl/text.rs:34-260 = the dense Qwen3 decoder (QK-norm before RoPE)
Rotary positions: the reference's dead code applies a plain 1-D RoPE at `seqlen_offsets`; HF Qwen3VL (whose checkpoints this
path loads, and the only runnable implementation here) uses the index-interleaved 3-axis MRoPE with sections [24, 20, 20] --
the same rule the reference implements for Qwen 3.5 (qwen3_5/modeling.rs:156-245).  This oracle follows HF (golden:
tests/golden/make_golden_qwen3_vl.py); for text-only prompts the two coincide (T = H = W).
"""
from typing import Dict, List, Optional, Sequence

import numpy as np

from oracle.qwen3_5_vision_oracle import VisionOracle, build_position_ids, gelu_erf, gelu_tanh, layer_norm, splice_image_features
from oracle.qwen3_oracle import Qwen3Config, Qwen3Oracle, rms_norm

F32 = np.float32


class DeepstackVisionOracle(VisionOracle):
    """Qwen3VLVisionModel::forward -> (merged tokens, [deepstack features]) (vision.rs:558-585): after every block listed in
    `deepstack_visual_indexes` a PatchMerger with use_postshuffle_norm = true (LayerNorm over the 4 x hidden regrouped row,
    vision.rs:236-276) turns the current hidden states into one [N / 4, out_hidden] feature map."""

    def forward_with_deepstack(self, pixel_values: np.ndarray, grid: Sequence[Sequence[int]]):
        idx = list(self.c.get("deepstack_visual_indexes", []))
        feats: List[Optional[np.ndarray]] = [None] * len(idx)
        w, p = self.w, self.p
        MH = self.hidden * self.merge ** 2

        def tap(li, x):
            if li in idx:
                k = idx.index(li)
                mp = f"{p}deepstack_merger_list.{k}."
                xn = layer_norm(x.reshape(-1, MH), w[mp + "norm.weight"], w[mp + "norm.bias"])
                hmid = self.merger_act((xn @ w[mp + "linear_fc1.weight"].T + w[mp + "linear_fc1.bias"]).astype(F32))
                feats[k] = (hmid @ w[mp + "linear_fc2.weight"].T + w[mp + "linear_fc2.bias"]).astype(F32)

        out = self.forward(pixel_values, grid, block_hook=tap)
        return out, feats


def mrope_rows(cos_table: np.ndarray, sin_table: np.ndarray, pos3: np.ndarray, section=(24, 20, 20)):
    """cos / sin rows [S, D/2] of the index-interleaved MRoPE: frequency i takes the H position when i % 3 == 1 and
    i < 3 * section[1], the W position when i % 3 == 2 and i < 3 * section[2], else the T position."""
    half = cos_table.shape[1]
    i = np.arange(half)
    axis = np.where((i % 3 == 1) & (i < 3 * section[1]), 1, np.where((i % 3 == 2) & (i < 3 * section[2]), 2, 0))
    rows = np.asarray(pos3, dtype=np.int64)[axis, :].T                   # [S, half]
    return cos_table[rows, i[None, :]], sin_table[rows, i[None, :]]


class Qwen3VLOracle:
    def __init__(self, cfg: dict, weights: Dict[str, np.ndarray], merger_gelu: str = "tanh", kv_dtype: str = "f32"):
        t = dict(cfg["text_config"])
        t["model_type"] = "qwen3"
        t["tie_word_embeddings"] = cfg.get("tie_word_embeddings", t.get("tie_word_embeddings", False))
        rp = t.get("rope_parameters") or t.get("rope_scaling") or {}
        t["rope_theta"] = rp.get("rope_theta", t.get("rope_theta", 5e6))
        self.section = tuple(rp.get("mrope_section", (24, 20, 20)))
        text_w = {k.replace("model.language_model.", "model."): v for k, v in weights.items() if not k.startswith("model.visual.")}
        self.text = Qwen3Oracle(Qwen3Config.from_json(t), text_w, kv_dtype=kv_dtype, max_pos=4096)
        self.vision = DeepstackVisionOracle(cfg["vision_config"], weights, merger_gelu=merger_gelu)
        self.image_token = cfg["image_token_id"]
        self.merge = cfg["vision_config"].get("spatial_merge_size", 2)
        self.next_pos = 0

    def prefill(self, ids: Sequence[int], pixel_values: np.ndarray, grid):
        if len(grid) == 0:                                              # text-only prompt
            feat, deep = np.zeros((0, self.text.cfg.hidden_size), F32), []
        else:
            feat, deep = self.vision.forward_with_deepstack(pixel_values, grid)
        pos3, nxt = build_position_ids(ids, grid, self.image_token, self.merge)
        emb = splice_image_features(ids, self.text.embed[np.array(ids)], feat, self.image_token)
        vis = np.array([t == self.image_token for t in ids])

        def inject(li, h):                                               # deepstack_process (text.rs:280-333)
            if li < len(deep):
                h = h.copy()
                h[vis] += deep[li]
            return h

        self.text.clear_kv_cache()
        self.text._rope_override = mrope_rows(self.text.cos, self.text.sin, pos3, self.section)
        try:
            h = self.text.forward_hidden(ids, 0, embeds=emb, after_layer=inject)
        finally:
            self.text._rope_override = None
        self.next_pos, self.len = nxt, len(ids)
        last = rms_norm(h[-1:], self.text.norm, self.text.cfg.rms_norm_eps)
        return feat, deep, self.text._mm(last, -1, "lm_head").astype(F32)[0]

    def decode(self, token: int):
        p = np.array([[self.next_pos]] * 3)
        self.text._rope_override = mrope_rows(self.text.cos, self.text.sin, p, self.section)
        try:
            lg = self.text.forward([token], self.len)
        finally:
            self.text._rope_override = None
        self.next_pos += 1
        self.len += 1
        return lg
