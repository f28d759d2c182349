"""CPU oracle: Qwen 3.5 / 3.6 / 3.8 hybrid decoder (Gated Delta Net + gated softmax attention),
restated from the reference in numpy f32.  TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

Follows (paths under /root/reference/crane-core/src):
  models/qwen3_5/config.rs:47-254      (text config, layer_types from full_attention_interval)
  models/qwen3_5/modeling.rs:45-79     Qwen35RmsNorm  x/rms * (1 + w)
  models/qwen3_5/modeling.rs:98-279    MRotaryEmbedding (inv_freq computed in **f32**), apply_mrope
                                       (first rot_dim dims only, rotate-half inside the slice, f32)
  models/qwen3_5/modeling.rs:413-564   FullAttention: q_proj -> per-head [q | gate], QK-norm (1+w),
                                       partial rope, GQA softmax, y * sigmoid(gate), o_proj
  models/qwen3_5/modeling.rs:622-628   Mlp: silu(gate) * up -> down
  models/qwen3_5/modeling.rs:784-832   DecoderLayer
  models/qwen3_5/model.rs:395-510      embed -> layers -> head(last position) -> [V]
  models/qwen3_5/prefill.rs:119-136    causal mask with -inf
  ops/gdn/layer.rs:122-238             GatedDeltaNet::forward, split_qkv, repeat_kv_heads (Interleaved)
  ops/gdn/conv.rs:23-133               causal depthwise conv1d (k=4) + SiLU, rolling state
  ops/gdn/backend.rs:26-71,90-156,169-211   l2_norm (eps 1e-6), recurrence, beta / g gates
  ops/gdn/norm.rs:39-45                RmsNormGated: rms_norm(y, w) * silu(z)   (plain w!)

Text-only path: the three MRoPE position components are equal, so cos/sin are plain table rows.
Pinned by HF transformers ``Qwen3_5ForCausalLM`` on identical synthetic weights
(tests/golden/qwen3_5_*.npz) -- HF is what the reference claims bit-exact arg-max parity with
(reference README.md:402-404) -- and by the reference's own KATs replayed in tests/test_oracle_kat.py and tests/test_qwen3_5.py; the
recurrence `gated_delta_rule` is additionally pinned on the reference's fused GPU kernel itself (kernels/cuda/gdn.cu built
into oracle/_ref/ by oracle/build_ref.sh; tests/test_gpu_ref_kernels.py, 2e-5).
Parity unpinned: matmul/softmax summation order, arg-max tie-break (as for qwen3).
"""
from __future__ import annotations

import json
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

from .qwen3_oracle import bf16_round, f16_round, silu, softmax_last

F32 = np.float32


@dataclass
class Qwen35Config:
    vocab_size: int
    hidden_size: int
    intermediate_size: int
    num_hidden_layers: int
    num_attention_heads: int
    num_key_value_heads: int
    head_dim: int
    max_position_embeddings: int
    rms_norm_eps: float
    linear_key_head_dim: int
    linear_value_head_dim: int
    linear_num_key_heads: int
    linear_num_value_heads: int
    linear_conv_kernel_dim: int = 4
    full_attention_interval: int = 4
    rope_theta: float = 1e7
    partial_rotary_factor: float = 0.25
    tie_word_embeddings: bool = False
    attn_output_gate: bool = True

    @classmethod
    def from_json(cls, d) -> "Qwen35Config":
        d = json.loads(d) if isinstance(d, str) else dict(d)
        if "text_config" in d and "hidden_size" not in d:
            tie = d.get("tie_word_embeddings", False)
            d = dict(d["text_config"])
            d.setdefault("tie_word_embeddings", tie)
        rp = d.get("rope_parameters") or {}
        kw = {k: d[k] for k in cls.__dataclass_fields__ if k in d}
        kw.setdefault("rope_theta", rp.get("rope_theta", 1e7))
        kw.setdefault("partial_rotary_factor", rp.get("partial_rotary_factor", 0.25))
        return cls(**kw)

    @property
    def rot_dim(self) -> int:                      # config.rs:226-229
        return int(self.head_dim * self.partial_rotary_factor)

    def layer_is_full(self, i: int) -> bool:       # config.rs:231-241
        return (i + 1) % self.full_attention_interval == 0

    @property
    def key_dim(self): return self.linear_num_key_heads * self.linear_key_head_dim
    @property
    def value_dim(self): return self.linear_num_value_heads * self.linear_value_head_dim
    @property
    def conv_dim(self): return 2 * self.key_dim + self.value_dim


def rms_norm_1p(x, w, eps):
    """Qwen35RmsNorm (modeling.rs:45-79): x / sqrt(mean(x^2)+eps) * (1 + w)."""
    x = x.astype(F32)
    ms = np.mean(x * x, axis=-1, keepdims=True, dtype=F32)
    return (x / np.sqrt(ms + F32(eps)) * (F32(1.0) + w)).astype(F32)


def rms_norm_plain(x, w, eps):
    x = x.astype(F32)
    ms = np.mean(x * x, axis=-1, keepdims=True, dtype=F32)
    return (x / np.sqrt(ms + F32(eps)) * w).astype(F32)


def mrope_tables(rot_dim: int, max_pos: int, theta: float):
    """MRotaryEmbedding::new (modeling.rs:106-124): base and exponent in f32."""
    half = rot_dim // 2
    base = F32(theta)
    inv = np.array([F32(1.0) / np.power(base, F32(i) * F32(2.0) / F32(rot_dim), dtype=F32) for i in range(half)], dtype=F32)
    pos = np.arange(max_pos, dtype=F32)
    fr = (pos[:, None] * inv[None, :]).astype(F32)
    return np.cos(fr).astype(F32), np.sin(fr).astype(F32)


def apply_partial_rope(x, cos, sin, rot_dim):
    """apply_mrope (modeling.rs:263-279) on [S, H, D]: rotate-half inside x[..., :rot_dim], pass the rest."""
    r2 = rot_dim // 2
    x1, x2, rest = x[..., :r2], x[..., r2:rot_dim], x[..., rot_dim:]
    c, s = cos[:, None, :], sin[:, None, :]
    return np.concatenate([x1 * c - x2 * s, x1 * s + x2 * c, rest], axis=-1).astype(F32)


def sigmoid(x):
    return (F32(1.0) / (F32(1.0) + np.exp(-x))).astype(F32)


def softplus(x):
    """ops/gdn/backend.rs:73-75: log(1 + exp(x))."""
    return np.log(F32(1.0) + np.exp(x)).astype(F32)


def l2_norm(x, eps=1e-6):
    """backend.rs:26-56: x / sqrt(sum(x^2) + eps)."""
    return (x / np.sqrt(np.sum(x * x, axis=-1, keepdims=True, dtype=F32) + F32(eps))).astype(F32)


def gated_delta_rule(q, k, v, g, beta, state):
    """gated_delta_rule_recurrence (backend.rs:90-156).  q,k [S,NV,K] (q already l2-normed, scaled here),
    v [S,NV,V], g,beta [S,NV], state [NV,K,V] f32 (updated in place).  Returns y [S,NV,V]."""
    S, NV, K = q.shape
    q = q * F32(1.0 / math.sqrt(K))
    ys = np.zeros_like(v, dtype=F32)
    for t in range(S):
        state *= np.exp(g[t])[:, None, None]
        kv = np.einsum("hkv,hk->hv", state, k[t]).astype(F32)
        delta = ((v[t] - kv) * beta[t][:, None]).astype(F32)
        state += k[t][:, :, None] * delta[:, None, :]
        ys[t] = np.einsum("hkv,hk->hv", state, q[t])
    return ys


class Qwen35Oracle:
    def __init__(self, cfg: Qwen35Config, weights: Dict[str, np.ndarray], kv_dtype: str = "f32",
                 max_pos: Optional[int] = None, prefix: str = "model."):
        self.cfg = c = cfg
        self.kv_dtype = kv_dtype
        w = {k: np.ascontiguousarray(v, dtype=F32) for k, v in weights.items()}
        self.w, self.p = w, prefix
        self.embed = w[prefix + "embed_tokens.weight"]
        # lm_head probed at root, else tied (model.rs:106-123)
        self.lm_head = w["lm_head.weight"] if ("lm_head.weight" in w and not c.tie_word_embeddings) else self.embed
        self.norm = w[prefix + "norm.weight"]
        self.cos, self.sin = mrope_tables(c.rot_dim, max_pos or c.max_position_embeddings, c.rope_theta)
        self.allreduce = lambda t: t      # tensor-parallel tests plug a gloo all-reduce in here (row-parallel partial sums)
        self.clear_kv_cache()

    def _lin(self, x, name, mixer=False):
        """One linear of the model: x @ W^T.  (oracle/qhybrid_oracle.py overrides it with ggml's quantised-activation arithmetic;
        `mixer`: x is the output of the attention / Gated-Delta-Net token mixer -- only that override cares.)"""
        return x @ self.w[name].T

    def clear_kv_cache(self):
        c = self.cfg
        self.kc = [None] * c.num_hidden_layers
        self.vc = [None] * c.num_hidden_layers
        self.conv = [np.zeros((c.conv_dim, c.linear_conv_kernel_dim - 1), F32) for _ in range(c.num_hidden_layers)]
        self.state = [np.zeros((c.linear_num_value_heads, c.linear_key_head_dim, c.linear_value_head_dim), F32)
                      for _ in range(c.num_hidden_layers)]

    # ---- FullAttention::forward (modeling.rs:413-564) ----
    def _full_attn(self, li, x, start):
        c, w = self.cfg, self.w
        p = f"{self.p}layers.{li}.self_attn."
        S, Hq, Hkv, D = x.shape[0], c.num_attention_heads, c.num_key_value_heads, c.head_dim
        qo = self._lin(x, p + "q_proj.weight").reshape(S, Hq, 2 * D)            # per-head [q | gate] (:428-455)
        q, gate = qo[..., :D], qo[..., D:].reshape(S, Hq * D)
        k = self._lin(x, p + "k_proj.weight").reshape(S, Hkv, D)
        v = self._lin(x, p + "v_proj.weight").reshape(S, Hkv, D)
        q = rms_norm_1p(q, w[p + "q_norm.weight"], c.rms_norm_eps)          # (:464-465)
        k = rms_norm_1p(k, w[p + "k_norm.weight"], c.rms_norm_eps)
        cos, sin = self.cos[start:start + S], self.sin[start:start + S]
        if getattr(self, "_rope_override", None) is not None:
            cos, sin = self._rope_override
        q = apply_partial_rope(q, cos, sin, c.rot_dim)
        k = apply_partial_rope(k, cos, sin, c.rot_dim)
        k, v = k.transpose(1, 0, 2), v.transpose(1, 0, 2)
        if self.kv_dtype == "bf16":
            k, v = bf16_round(k), bf16_round(v)
        elif self.kv_dtype == "f16":                  # CM_KV_F16 pages: IEEE binary16, RNE, saturating at +-65504
            k, v = f16_round(k), f16_round(v)
        elif self.kv_dtype in ("int8", "int4"):       # KvCache::Quant (qwen3_5/kv_cache.rs:209-342)
            from oracle.kv_quant_oracle import roundtrip
            bits = 8 if self.kv_dtype == "int8" else 4
            k, v = roundtrip(k, bits), roundtrip(v, bits)
        if self.kc[li] is None or start == 0:
            self.kc[li], self.vc[li] = k, v
        else:
            self.kc[li] = np.concatenate([self.kc[li][:, :start], k], axis=1)
            self.vc[li] = np.concatenate([self.vc[li][:, :start], v], axis=1)
        K, V = self.kc[li], self.vc[li]
        L, n_rep = K.shape[1], Hq // Hkv
        qh = q.transpose(1, 0, 2).reshape(Hkv, n_rep, S, D)
        sc = np.einsum("grsd,gld->grsl", qh, K, optimize=True).astype(F32) * F32(1.0 / math.sqrt(D))
        if S > 1:
            sc = np.where(np.arange(L)[None, :] <= (start + np.arange(S))[:, None], sc, F32(-np.inf))
        y = np.einsum("grsl,gld->grsd", softmax_last(sc), V, optimize=True).astype(F32)
        y = y.reshape(Hq, S, D).transpose(1, 0, 2).reshape(S, Hq * D)
        y = y * sigmoid(gate)                                              # (:516-522)
        return self._lin(y.astype(F32), p + "o_proj.weight", mixer=True).astype(F32)

    # ---- GatedDeltaNet::forward (ops/gdn/layer.rs:122-182) ----
    def _gdn(self, li, x):
        c, w = self.cfg, self.w
        p = f"{self.p}layers.{li}.linear_attn."
        S = x.shape[0]
        NK, NV, K, V = c.linear_num_key_heads, c.linear_num_value_heads, c.linear_key_head_dim, c.linear_value_head_dim
        mixed = self._lin(x, p + "in_proj_qkv.weight").astype(F32)             # [S, conv_dim]
        z = self._lin(x, p + "in_proj_z.weight").astype(F32)                   # [S, VD]
        b = (x @ w[p + "in_proj_b.weight"].T).astype(F32)                   # [S, NV]
        a = (x @ w[p + "in_proj_a.weight"].T).astype(F32)
        # causal depthwise conv1d: output t reads the window ending at t (conv.rs:47-57)
        ker = c.linear_conv_kernel_dim
        cw = w[p + "conv1d.weight"].reshape(c.conv_dim, ker)
        hidden = np.concatenate([self.conv[li], mixed.T], axis=1)           # [conv_dim, ker-1+S]
        self.conv[li] = hidden[:, -(ker - 1):].copy()
        out = np.zeros((c.conv_dim, S), F32)
        for j in range(ker):
            out += hidden[:, j:j + S] * cw[:, j:j + 1]
        mixed = silu(out).T                                                 # [S, conv_dim]
        q = mixed[:, :c.key_dim].reshape(S, NK, K)
        k = mixed[:, c.key_dim:2 * c.key_dim].reshape(S, NK, K)
        v = mixed[:, 2 * c.key_dim:].reshape(S, NV, V)
        vpg = NV // NK
        if vpg > 1:                                                         # Interleaved (HF) order (layer.rs:194-238)
            q = np.repeat(q, vpg, axis=1)
            k = np.repeat(k, vpg, axis=1)
        q, k = l2_norm(q), l2_norm(k)
        beta = sigmoid(b)
        g = (-np.exp(w[p + "A_log"]) * softplus(a + w[p + "dt_bias"])).astype(F32)   # backend.rs:197-211
        y = gated_delta_rule(q, k, v, g, beta, self.state[li])             # [S, NV, V]
        yn = rms_norm_plain(y.reshape(-1, V), w[p + "norm.weight"], c.rms_norm_eps) * silu(z.reshape(-1, V))
        return self._lin(yn.reshape(S, NV * V).astype(F32), p + "out_proj.weight", mixer=True).astype(F32)

    def mrope_cos_sin(self, pos3: np.ndarray, mrope_section=(11, 11, 10)):
        """cos_sin_with_position_ids (modeling.rs:156-245): INDEX-interleaved MRoPE -- column i of the
        half-rot table is served by axis i % 3 (T, H, W) until that axis' section runs out, else by T."""
        half = self.cfg.rot_dim // 2
        axis_of = np.zeros(half, dtype=np.int64)
        for dim, offset in ((1, 1), (2, 2)):
            limit = min(mrope_section[dim] * 3, half)
            axis_of[offset:limit:3] = dim
        cols = np.arange(half)
        rows = np.asarray(pos3, dtype=np.int64)[axis_of, :].T            # [S, half]
        return self.cos[rows, cols[None, :]], self.sin[rows, cols[None, :]]

    def forward(self, input_ids: Sequence[int], start_pos: int, embeds: Optional[np.ndarray] = None,
                pos3: Optional[np.ndarray] = None) -> np.ndarray:
        """Logits [V] of the LAST position (model.rs:504-510).  `embeds` [S,H] replaces the embedding lookup and
        `pos3` [3,S] the rotary positions (forward_embeds, model.rs:430-462) -- KV/cache index stays start_pos."""
        c, w = self.cfg, self.w
        if start_pos == 0:
            self.clear_kv_cache()
        h = (self.embed[np.asarray(input_ids, dtype=np.int64)] if embeds is None else embeds).astype(F32)
        self._rope_override = None if pos3 is None else self.mrope_cos_sin(pos3)
        for li in range(c.num_hidden_layers):
            p = f"{self.p}layers.{li}."
            xn = rms_norm_1p(h, w[p + "input_layernorm.weight"], c.rms_norm_eps)
            h = h + self.allreduce(self._full_attn(li, xn, start_pos) if c.layer_is_full(li) else self._gdn(li, xn))
            xn = rms_norm_1p(h, w[p + "post_attention_layernorm.weight"], c.rms_norm_eps)
            gate = self._lin(xn, p + "mlp.gate_proj.weight")
            up = self._lin(xn, p + "mlp.up_proj.weight")
            h = h + self.allreduce(self._lin((silu(gate) * up).astype(F32), p + "mlp.down_proj.weight").astype(F32))
        last = rms_norm_1p(h[-1:], self.norm, c.rms_norm_eps)
        return self._head(last).astype(F32)[0]

    def _head(self, last):
        return last @ self.lm_head.T

    forward_step = forward

    def generate(self, input_ids: Sequence[int], max_new_tokens: int, eos_token_ids: Sequence[int] = ()):
        """Greedy branch of qwen3_5::Model::generate (model.rs:853-943): prompt at step 0, multi-id EOS."""
        toks = [int(t) for t in input_ids]
        for i in range(max_new_tokens):
            ctx = len(toks) if i == 0 else 1
            logits = self.forward(toks[len(toks) - ctx:], len(toks) - ctx)
            nxt = int(np.argmax(logits))
            toks.append(nxt)
            if nxt in eos_token_ids:
                break
        return toks
