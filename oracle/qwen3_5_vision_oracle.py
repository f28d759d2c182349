"""CPU oracle: Qwen 3.5-VL vision tower + VLM glue, restated from the reference in numpy f32.
TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

Follows (paths under /root/reference/crane-core/src/models/qwen3_5):
  vision.rs:13-60            PatchEmbed: Conv3d(k = [T,P,P], stride P, no bias) + bias == one linear over the
                             flattened patch row (channel, temporal, y, x) (conv3d_temporal_2.rs:25-76)
  vision.rs:62-84            VisionMlp: fc1 -> act (config hidden_act, gelu_pytorch_tanh) -> fc2, with biases
  vision.rs:86-106           rotate_half over the WHOLE head dim, q*cos + rotate_half(q)*sin, f32
  vision.rs:108-178          VisionAttention: qkv linear (+bias), per-frame full (bidirectional) softmax attention
                             in f32 over cu_seqlens windows, proj (+bias)
  vision.rs:180-233          VisionBlock: LayerNorm(eps 1e-6) -> attn -> +res -> LayerNorm -> mlp -> +res
  vision.rs:235-279          PatchMerger: LayerNorm(hidden) -> reshape [N/4, 4*hidden] -> fc1 -> gelu -> fc2
                             (`xs.gelu()`: candle's Tensor::gelu is the tanh approximation [external]; HF uses erf:
                             both are available here through `merger_gelu`)
  vision.rs:281-300          VisionRotaryEmbedding(dim = head_dim/2), theta 1e4, f32
  vision.rs:382-489          fast_pos_embed_interpolate: bilinear blend of 4 learned rows, then block-major order
  vision.rs:491-541          rot_pos_emb: (row, col) coordinates in merge-block-major order
  vision.rs:543-584          cu_seqlens per frame; forward
  vlm.rs:190-241             build_position_ids (3-axis MRoPE positions, next position = base + max(t,h,w))
  vlm.rs:250-301             forward (splice image features, forward_embeds), decode_step with a separate MRoPE counter
  vlm.rs:433-468             splice_image_features
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import numpy as np

F32 = np.float32


def layer_norm(x, w, b, eps=1e-6):
    x = x.astype(F32)
    mu = x.mean(-1, keepdims=True, dtype=F32)
    var = ((x - mu) ** 2).mean(-1, keepdims=True, dtype=F32)
    return ((x - mu) / np.sqrt(var + F32(eps)) * w + b).astype(F32)


def gelu_tanh(x):
    return (F32(0.5) * x * (F32(1) + np.tanh(F32(math.sqrt(2 / math.pi)) * (x + F32(0.044715) * x * x * x)))).astype(F32)


def gelu_erf(x):
    from math import erf
    return (F32(0.5) * x * (F32(1) + np.vectorize(erf)(x / math.sqrt(2)).astype(F32))).astype(F32)


def rotate_half(x):
    d = x.shape[-1] // 2
    return np.concatenate([-x[..., d:], x[..., :d]], axis=-1)


class VisionOracle:
    def __init__(self, vcfg: dict, w: Dict[str, np.ndarray], prefix: str = "model.visual.", merger_gelu: str = "tanh"):
        self.c, self.p = vcfg, prefix
        self.w = {k: np.ascontiguousarray(v, dtype=F32) for k, v in w.items() if k.startswith(prefix)}
        self.merge = vcfg.get("spatial_merge_size", 2)
        self.hidden, self.heads = vcfg["hidden_size"], vcfg["num_heads"]
        self.grid_side = int(round(math.sqrt(vcfg["num_position_embeddings"])))
        self.merger_act = gelu_tanh if merger_gelu == "tanh" else gelu_erf

    # vision.rs:370-489
    def pos_embed_interp(self, grid: Sequence[Sequence[int]]):
        idx = [[], [], [], []]
        wts = [[], [], [], []]
        side = self.grid_side
        for t, h, w in grid:
            def lin(n):
                if n == 1:
                    return np.zeros(1, F32)
                step = F32(side - 1) / F32(n - 1)
                return (np.arange(n, dtype=F32) * step).astype(F32)
            hv, wv = lin(h), lin(w)
            hf, wf = np.floor(hv).astype(int), np.floor(wv).astype(int)
            hc, wc = np.minimum(np.ceil(hv).astype(int), side - 1), np.minimum(np.ceil(wv).astype(int), side - 1)
            dh, dw = (hv - hf).astype(F32), (wv - wf).astype(F32)
            for i in range(h):
                for j in range(w):
                    idx[0].append(hf[i] * side + wf[j]); idx[1].append(hf[i] * side + wc[j])
                    idx[2].append(hc[i] * side + wf[j]); idx[3].append(hc[i] * side + wc[j])
                    wts[0].append((F32(1) - dh[i]) * (F32(1) - dw[j])); wts[1].append((F32(1) - dh[i]) * dw[j])
                    wts[2].append(dh[i] * (F32(1) - dw[j])); wts[3].append(dh[i] * dw[j])
        table = self.w[self.p + "pos_embed.weight"]
        pe = sum(table[np.array(idx[k])] * np.array(wts[k], F32)[:, None] for k in range(4)).astype(F32)
        out, start, m = [], 0, self.merge
        for t, h, w in grid:
            blk = np.tile(pe[start:start + h * w], (t, 1)).reshape(t, h // m, m, w // m, m, self.hidden)
            out.append(blk.transpose(0, 1, 3, 2, 4, 5).reshape(t * h * w, self.hidden))
            start += h * w
        return np.concatenate(out, 0)

    # vision.rs:491-541
    def rot_pos_emb(self, grid):
        hd = self.hidden // self.heads
        dim = hd // 2
        inv = np.array([F32(1) / np.power(F32(10000.0), F32(i) / F32(dim), dtype=F32) for i in range(0, dim, 2)], dtype=F32)
        max_hw = max(max(g[1], g[2]) for g in grid)
        table = (np.arange(max_hw, dtype=F32)[:, None] * inv[None, :]).astype(F32)
        rows, cols, m = [], [], self.merge
        for t, h, w in grid:
            base = [(br * m + ir, bc * m + ic) for br in range(h // m) for bc in range(w // m) for ir in range(m) for ic in range(m)]
            for _ in range(t):
                rows += [r for r, _ in base]; cols += [c for _, c in base]
        return np.concatenate([table[np.array(rows)], table[np.array(cols)]], axis=-1)      # [N, hd/2]

    def forward(self, pixel_values: np.ndarray, grid: Sequence[Sequence[int]], block_hook=None) -> np.ndarray:
        """pixel_values [N_patches, C*T*P*P] -> merged image tokens [N/merge^2, out_hidden] (vision.rs:558-584).
        `block_hook(li, x)` sees the hidden states after block li (the DeepStack taps, vision.rs:572-579)."""
        w, p = self.w, self.p
        pw = w[p + "patch_embed.proj.weight"].reshape(self.hidden, -1)
        x = (pixel_values.astype(F32) @ pw.T + w[p + "patch_embed.proj.bias"]).astype(F32)
        x = x + self.pos_embed_interp(grid)
        rot = self.rot_pos_emb(grid)
        emb = np.concatenate([rot, rot], axis=-1)
        cos, sin = np.cos(emb).astype(F32), np.sin(emb).astype(F32)
        cu = [0]
        for t, h, ww in grid:
            for _ in range(t):
                cu.append(cu[-1] + h * ww)
        H, hd = self.heads, self.hidden // self.heads
        act = gelu_tanh if self.c.get("hidden_act", "gelu_pytorch_tanh") == "gelu_pytorch_tanh" else gelu_erf
        for li in range(self.c["depth"]):
            b = f"{p}blocks.{li}."
            xn = layer_norm(x, w[b + "norm1.weight"], w[b + "norm1.bias"])
            qkv = (xn @ w[b + "attn.qkv.weight"].T + w[b + "attn.qkv.bias"]).reshape(-1, 3, H, hd)
            q, k, v = qkv[:, 0], qkv[:, 1], qkv[:, 2]
            q = q * cos[:, None, :] + rotate_half(q) * sin[:, None, :]
            k = k * cos[:, None, :] + rotate_half(k) * sin[:, None, :]
            outs = []
            for s0, s1 in zip(cu[:-1], cu[1:]):
                qq, kk, vv = q[s0:s1].transpose(1, 0, 2), k[s0:s1].transpose(1, 0, 2), v[s0:s1].transpose(1, 0, 2)
                sc = (qq @ kk.transpose(0, 2, 1)).astype(F32) / F32(math.sqrt(hd))
                sc = np.exp(sc - sc.max(-1, keepdims=True))
                pr = (sc / sc.sum(-1, keepdims=True, dtype=F32)).astype(F32)
                outs.append((pr @ vv).transpose(1, 0, 2).reshape(s1 - s0, H * hd))
            a = np.concatenate(outs, 0).astype(F32)
            x = x + (a @ w[b + "attn.proj.weight"].T + w[b + "attn.proj.bias"]).astype(F32)
            xn = layer_norm(x, w[b + "norm2.weight"], w[b + "norm2.bias"])
            hmid = act((xn @ w[b + "mlp.linear_fc1.weight"].T + w[b + "mlp.linear_fc1.bias"]).astype(F32))
            x = x + (hmid @ w[b + "mlp.linear_fc2.weight"].T + w[b + "mlp.linear_fc2.bias"]).astype(F32)
            if block_hook is not None:
                block_hook(li, x)
        mp = p + "merger."
        xn = layer_norm(x, w[mp + "norm.weight"], w[mp + "norm.bias"]).reshape(-1, self.hidden * self.merge ** 2)
        hmid = self.merger_act((xn @ w[mp + "linear_fc1.weight"].T + w[mp + "linear_fc1.bias"]).astype(F32))
        return (hmid @ w[mp + "linear_fc2.weight"].T + w[mp + "linear_fc2.bias"]).astype(F32)


def build_position_ids(ids: Sequence[int], grid, image_token_id: int, merge: int, start_pos: int = 0):
    """vlm.rs:190-241 -> (pos3 [3,S] int64, next_mrope_pos)."""
    S = len(ids)
    pos = np.zeros((3, S), dtype=np.int64)
    nxt, img, i = start_pos, 0, 0
    while i < S:
        if ids[i] != image_token_id:
            pos[:, i] = nxt; nxt += 1; i += 1
            continue
        gt, gh, gw = grid[img][0], grid[img][1] // merge, grid[img][2] // merge
        span, hw = gt * gh * gw, gh * gw
        assert i + span <= S, "not enough image placeholder tokens"
        for k in range(span):
            pos[:, i + k] = (nxt + k // hw, nxt + (k % hw) // gw, nxt + (k % hw) % gw)
        nxt += max(gt, gh, gw); i += span; img += 1
    return pos, nxt


def splice_image_features(ids: Sequence[int], embeds: np.ndarray, image_embeds: np.ndarray, image_token_id: int):
    """vlm.rs:433-468: image rows replace the embeddings at placeholder positions, in order."""
    out = embeds.copy()
    k = 0
    for s, t in enumerate(ids):
        if t == image_token_id:
            assert k < image_embeds.shape[0], "more image-placeholder positions than image embeddings"
            out[s] = image_embeds[k]; k += 1
    return out
