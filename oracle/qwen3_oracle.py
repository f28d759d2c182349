"""CPU oracle: Qwen3 dense decoder, restated from the reference (numpy, f32).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

Follows ``/root/reference/crane-core/src/models/qwen3/modeling.rs`` (cited as
``modeling.rs:L``), ``.../qwen3/model.rs`` (``model.rs:L``),
``.../modules/rotary.rs`` (``rotary.rs:L``) and
``.../modules/kv_cache.rs``.  Arithmetic that lives in candle 0.11 (not
vendored: rms_norm, rope_thd, softmax, matmul, cpu flash-attn,
LogitsProcessor arg-max) is restated from its published semantics and pinned
by the reference's own known-answer tests, replayed in
``tests/test_oracle_kat.py``, and by HF ``transformers`` on identical weights
(``tests/golden/`` + ``tests/golden/make_golden_qwen3.py``).

Parity status: rotary tables / rotate-half / rms_norm / causal semantics are
pinned by reference KATs; matmul accumulation order, softmax summation order
and arg-max tie-break are NOT pinned by any reference test ("parity unpinned"
for those three; we use f32 pairwise/BLAS sums, max-subtracted softmax and
first-max arg-max).

Everything is float32, which is what the reference's CPU device runs
(``crane-serve/src/lib.rs:432-458``: CPU default dtype F32; candle CPU has no
BF16 matmul, ``modeling.rs:1630-1631``).  ``kv_dtype="bf16"`` additionally
rounds K/V to bfloat16 when they enter the cache, which is what the reference
does on a BF16 device (cache kept in model dtype, ``kv_cache.rs:38-101``).
``act_dtype="bf16x2"``/``"bf16"`` emulate the rounding points of the MI355X
MFMA prefill path (activations split into 2 / 1 bf16 terms before each GEMM).
"""
from __future__ import annotations

import json
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------
# config  (modeling.rs:94-130)
# --------------------------------------------------------------------------
@dataclass
class Qwen3Config:
    vocab_size: int
    hidden_size: int
    intermediate_size: int
    num_hidden_layers: int
    num_attention_heads: int
    num_key_value_heads: int
    max_position_embeddings: int
    rms_norm_eps: float
    head_dim: Optional[int] = None          # default hidden/heads (modeling.rs:126-129)
    rope_theta: float = 1_000_000.0         # default_rope_theta (modeling.rs:107)
    attention_bias: bool = False
    use_qk_norm: bool = True                # default true (modeling.rs:111)
    tie_word_embeddings: bool = True        # default true (modeling.rs:113)
    eos_token_id: Optional[int] = None

    @property
    def hd(self) -> int:
        return self.head_dim if self.head_dim else self.hidden_size // self.num_attention_heads

    @classmethod
    def from_json(cls, text_or_dict) -> "Qwen3Config":
        d = json.loads(text_or_dict) if isinstance(text_or_dict, str) else dict(text_or_dict)
        keys = cls.__dataclass_fields__.keys()
        return cls(**{k: d[k] for k in keys if k in d})


# --------------------------------------------------------------------------
# bf16 helpers (round-to-nearest-even, same bit trick as the HIP side)
# --------------------------------------------------------------------------
def f16_round(x: np.ndarray) -> np.ndarray:
    """Rounding point of the device's CM_KV_F16 pages (v_cvt_f16_f32 after a clamp): IEEE binary16, RNE, subnormals kept."""
    return np.clip(np.asarray(x, dtype=np.float32), -65504.0, 65504.0).astype(np.float16).astype(np.float32)


def bf16_round(x: np.ndarray) -> np.ndarray:
    """f32 -> nearest-even bf16 -> f32 (finite inputs)."""
    u = np.ascontiguousarray(x, dtype=F32).view(np.uint32)
    r = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)
    return r.view(F32)


def bf16_split2(x: np.ndarray) -> np.ndarray:
    """hi + lo with hi = bf16(x), lo = bf16(x - hi): ~16 mantissa bits."""
    hi = bf16_round(x)
    return hi + bf16_round(x - hi)


def _act(x: np.ndarray, mode: str) -> np.ndarray:
    if mode == "f32":
        return x
    if mode == "bf16":
        return bf16_round(x)
    if mode == "bf16x2":
        return bf16_split2(x)
    raise ValueError(mode)


# --------------------------------------------------------------------------
# rotary tables  (rotary.rs:29-46): inv_freq in f64 -> f32, freqs f32
# --------------------------------------------------------------------------
def rotary_tables(dim: int, max_pos: int, theta: float):
    inv = np.array([1.0 / (float(theta) ** (i / dim)) for i in range(0, dim, 2)],
                   dtype=np.float64).astype(F32)
    pos = np.arange(max_pos, dtype=F32)
    freqs = (pos[:, None] * inv[None, :]).astype(F32)     # f32 product, like the 1-col matmul
    return np.cos(freqs).astype(F32), np.sin(freqs).astype(F32)


def rope_thd(x: np.ndarray, cos: np.ndarray, sin: np.ndarray) -> np.ndarray:
    """candle_nn::rotary_emb::rope_thd on [S, H, D] (modeling.rs:358-359).

    Rotate-half pairing i <-> i + D/2 (pinned by rotary.rs:372-409):
      out[i]       = x[i]*cos[i] - x[i+D/2]*sin[i]
      out[i+D/2]   = x[i]*sin[i] + x[i+D/2]*cos[i]
    cos/sin: [S, D/2].
    """
    d2 = x.shape[-1] // 2
    x1, x2 = x[..., :d2], x[..., d2:]
    c, s = cos[:, None, :], sin[:, None, :]
    return np.concatenate([x1 * c - x2 * s, x1 * s + x2 * c], axis=-1).astype(F32)


def rms_norm(x: np.ndarray, w: np.ndarray, eps: float) -> np.ndarray:
    """candle_nn::rms_norm, plain weight (modeling.rs:660-669): x/sqrt(mean(x^2)+eps)*w."""
    x = x.astype(F32)
    ms = np.mean(x * x, axis=-1, keepdims=True, dtype=F32)
    return (x / np.sqrt(ms + F32(eps)) * w).astype(F32)


def silu(x: np.ndarray) -> np.ndarray:
    """x / (1 + exp(-x))  (kernels/cuda/fused_ops.cu:47-49)."""
    return (x / (F32(1.0) + np.exp(-x))).astype(F32)


def softmax_last(x: np.ndarray) -> np.ndarray:
    m = np.max(x, axis=-1, keepdims=True)
    e = np.exp(x - m)
    return (e / np.sum(e, axis=-1, keepdims=True, dtype=F32)).astype(F32)


# --------------------------------------------------------------------------
# model
# --------------------------------------------------------------------------
class Qwen3Oracle:
    """`Qwen3Model` (modeling.rs:720-1036) + `Model::generate` (model.rs:270-349).

    ``weights``: HF-named tensors as float32 numpy arrays.
    """

    def __init__(self, cfg: Qwen3Config, weights: Dict[str, np.ndarray],
                 kv_dtype: str = "f32", act_dtype: str = "f32",
                 max_pos: Optional[int] = None):
        self.cfg = cfg
        self.kv_dtype = kv_dtype
        self.act_dtype = act_dtype
        w = {k: np.ascontiguousarray(v, dtype=F32) for k, v in weights.items()}
        self.embed = w["model.embed_tokens.weight"]
        # tied lm_head shares the embedding tensor (modeling.rs:786-794)
        self.lm_head = self.embed if cfg.tie_word_embeddings else w["lm_head.weight"]
        self.norm = w["model.norm.weight"]
        self.layers = []
        for i in range(cfg.num_hidden_layers):
            p = f"model.layers.{i}."
            # merged QKV = cat(q,k,v) rows (modeling.rs:187-204); gate||up (modeling.rs:582-588)
            lw = dict(
                qkv=np.concatenate([w[p + "self_attn.q_proj.weight"],
                                    w[p + "self_attn.k_proj.weight"],
                                    w[p + "self_attn.v_proj.weight"]], axis=0),
                o=w[p + "self_attn.o_proj.weight"],
                q_norm=w.get(p + "self_attn.q_norm.weight") if cfg.use_qk_norm else None,
                k_norm=w.get(p + "self_attn.k_norm.weight") if cfg.use_qk_norm else None,
                gate_up=np.concatenate([w[p + "mlp.gate_proj.weight"],
                                        w[p + "mlp.up_proj.weight"]], axis=0),
                down=w[p + "mlp.down_proj.weight"],
                ln1=w[p + "input_layernorm.weight"],
                ln2=w[p + "post_attention_layernorm.weight"],
            )
            self.layers.append(lw)
        mp = max_pos or cfg.max_position_embeddings
        self.cos, self.sin = rotary_tables(cfg.hd, mp, cfg.rope_theta)
        self.clear_kv_cache()

    # -- linear: x @ W^T, or a quantised mat-vec with ggml's quantised-activation semantics when `qmats` maps
    #    (layer, name) -> [QuantMatrix, ...] whose outputs are concatenated (q|k|v, gate|up) -----------------
    qmats = None

    def _mm(self, x, li, name):
        if self.qmats is not None and (li, name) in self.qmats:
            return np.concatenate([qm.vecdot(x) for qm in self.qmats[(li, name)]], axis=-1).astype(F32)
        W = self.lm_head if name == "lm_head" else self.layers[li][name]
        return x @ W.T

    # -- KV cache (kv_cache.rs:38-101: contiguous BHSD, append) ------------
    def clear_kv_cache(self):
        self.k_cache: List[Optional[np.ndarray]] = [None] * self.cfg.num_hidden_layers
        self.v_cache: List[Optional[np.ndarray]] = [None] * self.cfg.num_hidden_layers
        self.cache_len = 0

    def _append_kv(self, li: int, k: np.ndarray, v: np.ndarray, start_pos: int):
        # k, v: [Hkv, S, D]
        if self.kv_dtype == "bf16":
            k, v = bf16_round(k), bf16_round(v)
        elif self.kv_dtype == "f16":                  # CM_KV_F16 pages: IEEE binary16, RNE, saturating at +-65504
            k, v = f16_round(k), f16_round(v)
        elif self.kv_dtype in ("int8", "int4"):       # KvCache::Quant (qwen3_5/kv_cache.rs:209-342)
            from oracle.kv_quant_oracle import roundtrip
            bits = 8 if self.kv_dtype == "int8" else 4
            k, v = roundtrip(k, bits), roundtrip(v, bits)
        if self.k_cache[li] is None or start_pos == 0:
            self.k_cache[li], self.v_cache[li] = k, v
        else:
            self.k_cache[li] = np.concatenate([self.k_cache[li][:, :start_pos], k], axis=1)
            self.v_cache[li] = np.concatenate([self.v_cache[li][:, :start_pos], v], axis=1)
        return self.k_cache[li], self.v_cache[li]

    # -- Attention::forward (modeling.rs:307-533) ---------------------------
    def _attention(self, li: int, x: np.ndarray, start_pos: int) -> np.ndarray:
        cfg, lw = self.cfg, self.layers[li]
        S = x.shape[0]
        Hq, Hkv, D = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.hd
        qkv = self._mm(_act(x, self.act_dtype), li, "qkv")                # modeling.rs:318-323
        q = qkv[:, :Hq * D].reshape(S, Hq, D)                           # BSHD (modeling.rs:335-339)
        k = qkv[:, Hq * D:(Hq + Hkv) * D].reshape(S, Hkv, D)
        v = qkv[:, (Hq + Hkv) * D:].reshape(S, Hkv, D)
        if lw["q_norm"] is not None:                                    # QK-norm BEFORE rope (:341-353)
            q = rms_norm(q, lw["q_norm"], cfg.rms_norm_eps)
            k = rms_norm(k, lw["k_norm"], cfg.rms_norm_eps)
        cos, sin = self.cos[start_pos:start_pos + S], self.sin[start_pos:start_pos + S]
        if self._rope_override is not None:                             # multimodal rotary rows (oracle/qwen3_vl_oracle.py)
            cos, sin = self._rope_override
        q = rope_thd(q, cos, sin)                                       # :358-359
        k = rope_thd(k, cos, sin)
        K, V = self._append_kv(li, k.transpose(1, 0, 2), v.transpose(1, 0, 2), start_pos)  # :362-366
        L = K.shape[1]
        n_rep = Hq // Hkv
        scale = F32(1.0 / math.sqrt(D))
        # causal with kv_offset = L - S (modeling.rs:436-438); GQA by integer division
        qh = q.transpose(1, 0, 2).reshape(Hkv, n_rep, S, D)
        scores = np.einsum("grsd,gld->grsl", qh, K, optimize=True).astype(F32) * scale
        if S > 1:
            qpos = start_pos + np.arange(S)[:, None]
            kpos = np.arange(L)[None, :]
            scores = np.where(kpos <= qpos, scores, F32(-np.inf))
        p = softmax_last(scores)
        out = np.einsum("grsl,gld->grsd", p, V, optimize=True).astype(F32)
        out = out.reshape(Hq, S, D).transpose(1, 0, 2).reshape(S, Hq * D)
        return self._mm(_act(out, self.act_dtype), li, "o").astype(F32)   # o_proj

    # -- Mlp::forward (modeling.rs:608-642) ---------------------------------
    def _mlp(self, li: int, x: np.ndarray) -> np.ndarray:
        lw, I = self.layers[li], self.cfg.intermediate_size
        gu = self._mm(_act(x, self.act_dtype), li, "gate_up")
        h = silu(gu[:, :I]) * gu[:, I:]
        return self._mm(_act(h, self.act_dtype), li, "down").astype(F32)

    # -- Qwen3Model::forward/decode (modeling.rs:942-953, 984-1036) ----------
    _rope_override = None       # (cos, sin) rows [S, D/2] replacing the table slice (multimodal rotary positions)

    def forward_hidden(self, input_ids: Sequence[int], start_pos: int, embeds: Optional[np.ndarray] = None,
                       after_layer=None) -> np.ndarray:
        """`embeds` [S, H] replaces the embedding rows (image features spliced in); `after_layer(li, h)` runs after each
        decoder layer (DeepStack injection) -- both only used by the vision-language oracle."""
        cfg = self.cfg
        ids = np.asarray(input_ids, dtype=np.int64)
        h = self.embed[ids].astype(F32) if embeds is None else np.asarray(embeds, dtype=F32)
        for li, lw in enumerate(self.layers):                           # DecoderLayer::forward :698-716
            h = h + self._attention(li, rms_norm(h, lw["ln1"], cfg.rms_norm_eps), start_pos)
            h = h + self._mlp(li, rms_norm(h, lw["ln2"], cfg.rms_norm_eps))
            if after_layer is not None:
                h = after_layer(li, h)
        self.cache_len = start_pos + len(ids)
        return h

    def forward(self, input_ids: Sequence[int], start_pos: int) -> np.ndarray:
        """Logits of the LAST position only, shape [vocab] (modeling.rs:1024-1035)."""
        h = self.forward_hidden(input_ids, start_pos)
        last = rms_norm(h[-1:], self.norm, self.cfg.rms_norm_eps)
        return self._mm(last, -1, "lm_head").astype(F32)[0]

    forward_step = forward      # ModelBackend::forward_step (backend.rs:41)

    # -- Model::generate, greedy branch (model.rs:275-349) -------------------
    def generate(self, input_ids: Sequence[int], max_new_tokens: int,
                 eos_token_id: Optional[int] = None,
                 repetition_penalty: float = 1.0, repeat_last_n: int = 5,
                 return_logits: bool = False):
        self.clear_kv_cache()
        tokens = list(int(t) for t in input_ids)
        all_logits = []
        for index in range(max_new_tokens):
            ctx = 1 if index > 0 else len(tokens)                       # whole prompt at step 0 (:299-302)
            start_pos = len(tokens) - ctx
            logits = self.forward(tokens[start_pos:], start_pos)
            if abs(repetition_penalty - 1.0) >= np.finfo(np.float32).eps:
                logits = apply_repeat_penalty(logits, repetition_penalty,
                                              tokens[max(0, len(tokens) - repeat_last_n):])
            if return_logits:
                all_logits.append(logits.copy())
            nxt = int(np.argmax(logits))                                # temperature None => arg-max
            tokens.append(nxt)
            if eos_token_id is not None and nxt == eos_token_id:
                break
        return (tokens, all_logits) if return_logits else tokens       # prompt ++ generated (:348)


def apply_repeat_penalty(logits: np.ndarray, penalty: float, context: Sequence[int]) -> np.ndarray:
    """candle_transformers::utils::apply_repeat_penalty (model.rs:306-315):
    each distinct context token once: logit >= 0 -> /penalty, else *penalty."""
    out = logits.astype(F32).copy()
    seen = set()
    for t in context:
        if t in seen or t >= out.shape[0]:
            continue
        seen.add(t)
        out[t] = out[t] / F32(penalty) if out[t] >= 0 else out[t] * F32(penalty)
    return out


# --------------------------------------------------------------------------
# stand-alone attention restatements used by the KAT tests
# --------------------------------------------------------------------------
def naive_gqa_attention(q, k, v, scale, causal_offset=None):
    """matmul+softmax reference of modeling.rs:1467-1735 tests. q [Hq,Sq,D]; k,v [Hkv,L,D]."""
    Hq, Sq, D = q.shape
    Hkv, L, _ = k.shape
    n_rep = Hq // Hkv
    out = np.zeros_like(q, dtype=F32)
    for h in range(Hq):
        s = (q[h] @ k[h // n_rep].T).astype(F32) * F32(scale)
        if causal_offset is not None:
            qpos = causal_offset + np.arange(Sq)[:, None]
            s = np.where(np.arange(L)[None, :] <= qpos, s, F32(-np.inf))
        out[h] = softmax_last(s) @ v[h // n_rep]
    return out


def flash_gqa_attention(q, k, v, scale, causal_offset=None):
    """Online-softmax single pass, the algorithm of candle's cpu flash_attn used by
    modules/flash_attn.rs:29-42 (f32 accumulators, GQA by integer division)."""
    Hq, Sq, D = q.shape
    Hkv, L, _ = k.shape
    n_rep = Hq // Hkv
    out = np.zeros((Hq, Sq, D), dtype=F32)
    for h in range(Hq):
        kh, vh = k[h // n_rep], v[h // n_rep]
        for i in range(Sq):
            m, l, acc = F32(-np.inf), F32(0), np.zeros(D, dtype=F32)
            hi = L if causal_offset is None else min(L, causal_offset + i + 1)
            for j in range(hi):
                s = F32(np.dot(q[h, i], kh[j])) * F32(scale)
                m_new = max(m, s)
                a = F32(np.exp(m - m_new)) if np.isfinite(m) else F32(0)
                p = F32(np.exp(s - m_new))
                acc = acc * a + p * vh[j]
                l = l * a + p
                m = m_new
            out[h, i] = acc / l
    return out
