"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- a decode ROUND of a group of sequences over Q8_0-layout weights with ggml's
quantised-activation semantics (`LinearLayer::Quantized`, crane-core/src/ops/linear.rs:18-51: candle's CPU QMatMul = the activation
row quantised to Q8_0 blocks, ggml_vec_dot_q8_0_q8_0 per output; ISQ through quantize_row_q8_0_ref / _q4_0_ref / _q5_0_ref,
ops/linear.rs:53-116), around the dense Qwen3 layer of oracle/qwen3_oracle.py (qwen3/modeling.rs:307-533, 608-642, 698-716).

Why a separate, TEACHER-FORCED oracle: an 8-bit activation code whose pre-rounding value sits within ~1e-5 of a .5 boundary rounds
one way or the other depending on the last bit of an f32 sum -- any two correct implementations of these semantics part ways at such
a code by a whole code step, and the difference (~1e-2 of the logit range end to end) says nothing about either.  So the device
reports the codes and block scales every projection consumed (cm_debug_set("q_capture")), and this oracle
  1. quantises ITS OWN activation rows with the reference arithmetic and compares: every code that differs must differ by exactly
     one step AND the oracle's own pre-rounding value must lie within `tie_tol` of the .5 boundary between the two codes (a block
     scale may differ by one f16 step when amax / 127 sits on an f16 rounding boundary); anything else is an error;
  2. continues from the DEVICE's codes (the verified alternative rounding), so that the outputs of the projection -- and the logits
     at the end -- must agree to f32 summation order, with no allowance for flips.
The arithmetic (quantisers, vec_dot) is oracle/c/q8_ref.c, pinned bit for bit on oracle/gguf_oracle.py (tests/test_gguf_oracle.py);
both are restatements of the published ggml algorithm -- PARITY UNPINNED against candle / ggml themselves (absent from the image).
"""
import ctypes as C
import math
from typing import Dict, List

import numpy as np

from crane_amd import synth
from oracle import c_oracle
from oracle.qwen3_oracle import F32, rms_norm, rope_thd, rotary_tables, silu, softmax_last

FMT = {"q8_0": 8, "q4_0": 2, "q5_0": 6}


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class QMat:
    """[N, K] matrix as int8 codes + f32 block scales (after the f16 round trip), quantised from f32 rows by the reference quantiser."""

    def __init__(self, lib, w: np.ndarray, fmt: int):
        w = np.ascontiguousarray(w, np.float32)
        self.N, self.K = w.shape
        self.q = np.empty((self.N, self.K), np.int8)
        self.d = np.empty((self.N, self.K // 32), np.float32)
        assert lib.qc_quantize_ref(fmt, _p(w, C.c_float), w.size, _p(self.q, C.c_int8), _p(self.d, C.c_float)) == 0


class Q8GroupOracle:
    def __init__(self, cfg: dict, isq: str, seed: int = 0, max_pos: int = 64):
        self.lib = c_oracle._lib()
        self.lib.qc_set_threads(c_oracle.host_threads())
        self.cfg = cfg
        fmt = FMT[isq]
        self.H, self.I = cfg["hidden_size"], cfg["intermediate_size"]
        self.Hq, self.Hkv = cfg["num_attention_heads"], cfg["num_key_value_heads"]
        self.D = cfg.get("head_dim") or self.H // self.Hq
        self.eps = cfg.get("rms_norm_eps", 1e-6)
        self.L = cfg["num_hidden_layers"]
        spec = {n: (shape, std, off) for n, shape, std, off in synth.specs_for(cfg)}

        def tensor(name):
            shape, std, off = spec[name]
            rows, cols = (shape[0], shape[1]) if len(shape) == 2 else (1, shape[0])
            out = np.empty((rows, cols), np.float32)
            self.lib.qc_synth_f32(name.encode(), seed, float(std), float(off), rows, cols, _p(out, C.c_float))
            return out if len(shape) == 2 else out[0]

        self.embed = tensor("model.embed_tokens.weight")
        self.norm = tensor("model.norm.weight")
        tied = cfg.get("tie_word_embeddings", True)
        # ISQ quantises every linear and an untied head; a tied table stays bf16 (qwen3_5/model.rs:617-626, model_factory.rs:482-536)
        self.head = None if tied else QMat(self.lib, tensor("lm_head.weight"), fmt)
        self.layers = []
        for i in range(self.L):
            p = f"model.layers.{i}."
            lw = dict(ln1=tensor(p + "input_layernorm.weight"), ln2=tensor(p + "post_attention_layernorm.weight"),
                      qn=tensor(p + "self_attn.q_norm.weight"), kn=tensor(p + "self_attn.k_norm.weight"))
            for k, n in (("q", "self_attn.q_proj"), ("k", "self_attn.k_proj"), ("v", "self_attn.v_proj"), ("o", "self_attn.o_proj"),
                         ("gate", "mlp.gate_proj"), ("up", "mlp.up_proj"), ("down", "mlp.down_proj")):
                lw[k] = QMat(self.lib, tensor(p + n + ".weight"), fmt)
            self.layers.append(lw)
        self.cos, self.sin = rotary_tables(self.D, max_pos, cfg.get("rope_theta", 1e6))
        self.kc: Dict[int, List] = {}
        self.stats = dict(codes=0, flipped=0, scales=0, scale_steps=0, worst_tie=0.0)

    # ---- quantise the rows like the reference, check the device's codes against them, continue from the device's ----
    def _quant(self, x: np.ndarray, cap, tie_tol: float):
        M, K = x.shape
        Kc, codes, scales = next(cap)
        assert Kc == K and codes.shape == (M, K), (Kc, K, codes.shape)
        xb = x.reshape(M, K // 32, 32)
        amax = np.abs(xb).max(axis=2)
        d = (amax / F32(127.0)).astype(F32)
        idv = np.where(d != 0, F32(1.0) / np.where(d != 0, d, 1), F32(0)).astype(F32)
        t = (xb * idv[:, :, None]).astype(F32).reshape(M, K)                      # the value roundf() sees
        t64 = t.astype(np.float64)
        mine = np.copysign(np.floor(np.abs(t64) + 0.5), t64).astype(np.int32)
        dev = codes.astype(np.int32)
        diff = dev != mine
        self.stats["codes"] += dev.size
        if diff.any():
            assert np.abs(dev - mine)[diff].max() == 1, "a device code differs from the reference rounding by more than one step"
            # distance of the oracle's own pre-rounding value from the boundary between the two candidate codes
            edge = (np.minimum(np.abs(dev), np.abs(mine)) + 0.5)[diff]
            gap = np.abs(np.abs(t64[diff]) - edge)
            same_sign = (np.sign(dev[diff]) * np.sign(mine[diff])) >= 0
            assert same_sign.all() and gap.max() < tie_tol, f"a differing code is not a rounding tie: gap {gap.max():.3e}"
            self.stats["flipped"] += int(diff.sum())
            self.stats["worst_tie"] = max(self.stats["worst_tie"], float(gap.max()))
        d16 = d.astype(np.float16).astype(F32)
        sd = scales != d16
        self.stats["scales"] += d16.size
        if sd.any():
            # one binary16 step apart, and only where amax / 127 itself sits next to a binary16 rounding boundary
            lo = np.nextafter(d16.astype(np.float16), np.float16(0)).astype(F32)
            hi = np.nextafter(d16.astype(np.float16), np.float16(np.inf)).astype(F32)
            ok = (scales == lo) | (scales == hi)
            assert ok[sd].all(), "a device block scale is not the f16 neighbour of the reference scale"
            mid = np.where(scales == lo, (d16 + lo) * F32(0.5), (d16 + hi) * F32(0.5))
            assert (np.abs(d - mid)[sd] <= np.abs(d)[sd] * 1e-5).all(), "a differing block scale is not an f16 rounding tie"
            self.stats["scale_steps"] += int(sd.sum())
        return np.ascontiguousarray(codes, np.int8), np.ascontiguousarray(scales, np.float32)

    def _mm(self, q: np.ndarray, d: np.ndarray, mats) -> np.ndarray:
        M = q.shape[0]
        outs = []
        for w in mats:
            o = np.empty((M, w.N), np.float32)
            assert self.lib.qc_vec_dot_q8_rows(_p(w.q, C.c_int8), _p(w.d, C.c_float), w.N, w.K, _p(q, C.c_int8), _p(d, C.c_float), M,
                                               _p(o, C.c_float)) == 0
            outs.append(o)
        return outs[0] if len(outs) == 1 else np.concatenate(outs, axis=1)

    def step(self, seq_ids, toks, captures, tie_tol: float = 2e-3) -> np.ndarray:
        """One decode round: sequence seq_ids[b] (its K/V kept here, f32) takes token toks[b] at its next position.
        `captures`: the device's q_capture records of the same round, [(K, codes [M, K] int8, scales [M, K / 32] f32), ...] in
        consumption order.  Returns the logits [M, V]."""
        cap = iter(captures)
        M, H, D, Hq, Hkv = len(toks), self.H, self.D, self.Hq, self.Hkv
        x = self.embed[np.asarray(toks, np.int64)].astype(F32)
        for s in seq_ids:
            self.kc.setdefault(s, [[None, None] for _ in range(self.L)])
        pos = [0 if self.kc[s][0][0] is None else self.kc[s][0][0].shape[1] for s in seq_ids]
        for li, lw in enumerate(self.layers):
            q8, d8 = self._quant(rms_norm(x, lw["ln1"], self.eps), cap, tie_tol)
            qkv = self._mm(q8, d8, (lw["q"], lw["k"], lw["v"]))
            attn = np.empty((M, Hq * D), F32)
            for b, s in enumerate(seq_ids):
                qh = qkv[b, :Hq * D].reshape(1, Hq, D)
                kh = qkv[b, Hq * D:(Hq + Hkv) * D].reshape(1, Hkv, D)
                vh = qkv[b, (Hq + Hkv) * D:].reshape(1, Hkv, D)
                qh = rms_norm(qh, lw["qn"], self.eps); kh = rms_norm(kh, lw["kn"], self.eps)      # QK-norm before RoPE (modeling.rs:341-353)
                c, sn = self.cos[pos[b]:pos[b] + 1], self.sin[pos[b]:pos[b] + 1]
                qh = rope_thd(qh, c, sn); kh = rope_thd(kh, c, sn)
                kv = self.kc[s][li]
                kv[0] = kh.transpose(1, 0, 2) if kv[0] is None else np.concatenate([kv[0], kh.transpose(1, 0, 2)], axis=1)
                kv[1] = vh.transpose(1, 0, 2) if kv[1] is None else np.concatenate([kv[1], vh.transpose(1, 0, 2)], axis=1)
                qg = qh[0].reshape(Hkv, Hq // Hkv, D)
                sc = np.einsum("grd,gld->grl", qg, kv[0]).astype(F32) * F32(1.0 / math.sqrt(D))
                attn[b] = np.einsum("grl,gld->grd", softmax_last(sc), kv[1]).astype(F32).reshape(Hq * D)
            q8, d8 = self._quant(attn, cap, tie_tol)
            x = (x + self._mm(q8, d8, (lw["o"],))).astype(F32)
            q8, d8 = self._quant(rms_norm(x, lw["ln2"], self.eps), cap, tie_tol)
            gu = self._mm(q8, d8, (lw["gate"], lw["up"]))
            h = (silu(gu[:, :self.I]) * gu[:, self.I:]).astype(F32)
            q8, d8 = self._quant(h, cap, tie_tol)
            x = (x + self._mm(q8, d8, (lw["down"],))).astype(F32)
        last = rms_norm(x, self.norm, self.eps)
        if self.head is None:
            return (last @ self.embed.T).astype(F32)
        q8, d8 = self._quant(last, cap, tie_tol)
        assert next(cap, None) is None, "the device quantised more activation rows than the round has projections"
        return self._mm(q8, d8, (self.head,))


def parse_captures(flat: np.ndarray):
    """cm_debug_read("q_capture") -> [(K, codes [M, K] int8, scales [M, K / 32] f32)]"""
    out, i = [], 0
    while i < flat.size:
        K, M = int(flat[i]), int(flat[i + 1]); i += 2
        codes = flat[i:i + M * K].astype(np.int8).reshape(M, K); i += M * K
        sc = flat[i:i + M * (K // 32)].astype(np.float32).reshape(M, K // 32); i += M * (K // 32)
        out.append((K, codes, sc))
    return out
