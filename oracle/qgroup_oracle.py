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
from oracle.qwen3_oracle import F32, f16_round, rms_norm, rope_thd, rotary_tables, silu, softmax_last

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
    def __init__(self, cfg: dict, isq: str, seed: int = 0, max_pos: int = 64, kv_dtype: str = "f32"):
        # kv_dtype "f16": K / V rows are rounded to IEEE binary16 (saturating) when they enter the cache -- the rounding point of the
        # device's default CM_KV_F16 pages (oracle/qwen3_oracle.py f16_round); "f32": the reference CPU cache
        assert kv_dtype in ("f32", "f16")
        self.kv_dtype = kv_dtype
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
    def _quant(self, x: np.ndarray, cap, tie_tol: float, scale_tol: float = 1e-5):
        M, K = x.shape
        self.n_quant = getattr(self, "n_quant", 0) + 1
        if cap is None:                      # free-running (CPU self-checks of this oracle): its own reference rounding
            x = np.ascontiguousarray(x, np.float32)
            q8 = np.empty((M, K), np.int8); d8 = np.empty((M, K // 32), np.float32)
            assert self.lib.qc_quantize_ref(FMT["q8_0"], _p(x, C.c_float), x.size, _p(q8, C.c_int8), _p(d8, C.c_float)) == 0
            return q8, d8
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
            assert same_sign.all() and gap.max() < tie_tol, f"a differing code is not a rounding tie: gap {gap.max():.3e} (record {self.n_quant}, K {K}, tol {tie_tol})"
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
            off = float((np.abs(d - mid)[sd] / np.abs(d)[sd]).max())
            assert off <= scale_tol, f"a differing block scale is not an f16 rounding tie: {off:.3e} (record {self.n_quant}, K {K}, tol {scale_tol})"
            self.stats["scale_steps"] += int(sd.sum())
        return np.ascontiguousarray(codes, np.int8), np.ascontiguousarray(scales, np.float32)

    def _quant_attn(self, x: np.ndarray, cap, rel_tol: float):
        """The o_proj input when the attention rows come from a matrix-core kernel with its own error budget (bf16 probabilities, 16-bit
        K / V and queries): a row x' within rel_tol x max|row| of the oracle's row x is as good as x, and its block maxima -- hence the
        block scales -- need not be f16 neighbours of the oracle's where a block is small against its row.  Checked instead: the
        device's blocks are a Q8_0 quantisation of SOME such x': |code * d - x| <= (1/2 + 127 * 2^-11) d + rel_tol * max|row| per
        element (half a code step, the f16 rounding of d at the largest code, the budget), codes in [-127, 127].  Then the device's codes."""
        M, K = x.shape
        self.n_quant = getattr(self, "n_quant", 0) + 1
        Kc, codes, scales = next(cap)
        assert Kc == K and codes.shape == (M, K), (Kc, K, codes.shape)
        assert np.abs(codes.astype(np.int32)).max() <= 127 and (scales >= 0).all()
        deq = (codes.reshape(M, K // 32, 32).astype(F32) * scales[:, :, None]).reshape(M, K)
        bound = np.repeat((0.5 + 127.0 / 2048.0) * scales, 32, axis=1) + rel_tol * np.abs(x).max(axis=1, keepdims=True)
        over = np.abs(deq - x) - bound
        assert over.max() <= 0, f"attention rows: a device block is not a Q8_0 quantisation of a row within {rel_tol} of the oracle's (record {self.n_quant}, excess {over.max():.3e})"
        self.stats["codes"] += codes.size
        self.stats["scales"] += scales.size
        self.stats["worst_attn"] = max(self.stats.get("worst_attn", 0.0),
                                       float((np.maximum(np.abs(deq - x) - np.repeat((0.5 + 127.0 / 2048.0) * scales, 32, axis=1), 0) / np.abs(x).max(axis=1, keepdims=True)).max()))
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

    def step(self, seq_ids, toks, captures, tie_tol: float = 2e-3, attn_tol: float = None) -> np.ndarray:
        """One decode round: sequence seq_ids[b] (its K/V kept here, f32) takes token toks[b] at its next position.
        `captures`: the device's q_capture records of the same round, [(K, codes [M, K] int8, scales [M, K / 32] f32), ...] in
        consumption order.  Returns the logits [M, V]."""
        cap = iter(captures) if captures is not None else None
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
                if self.kv_dtype == "f16":
                    kh, vh = f16_round(kh), f16_round(vh)
                kv[0] = kh.transpose(1, 0, 2) if kv[0] is None else np.concatenate([kv[0], kh.transpose(1, 0, 2)], axis=1)
                kv[1] = vh.transpose(1, 0, 2) if kv[1] is None else np.concatenate([kv[1], vh.transpose(1, 0, 2)], axis=1)
                qg = qh[0].reshape(Hkv, Hq // Hkv, D)
                sc = np.einsum("grd,gld->grl", qg, kv[0]).astype(F32) * F32(1.0 / math.sqrt(D))
                attn[b] = np.einsum("grl,gld->grd", softmax_last(sc), kv[1]).astype(F32).reshape(Hq * D)
            if attn_tol is None: q8, d8 = self._quant(attn, cap, tie_tol)
            else: q8, d8 = self._quant_attn(attn, cap, attn_tol)                           # (matrix-core attention: see _quant_attn)
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
        assert cap is None or next(cap, None) is None, "the device quantised more activation rows than the round has projections"
        return self._mm(q8, d8, (self.head,))


    def prefill(self, seq_ids, prompts, captures, tie_tol: float = 2e-3, panel: int = None, head_captured: bool = False,
                attn_tol: float = None):
        """One prompt pass over whole prompts of the (fresh or continued) sequences seq_ids, rows concatenated in order -- the device's
        Model::prefill_layers on the int8 matrix cores: every projection quantises ALL rows of the pass at once (panel = None; the
        kernel walks them in m-panels of 256 rows, which the arithmetic does not see), one capture record per projection input, in the
        device's order: per layer the input norm (qkv), the attention rows (o_proj), the post-attention norm (gate|up) and
        silu(gate) * up (down_proj).  (`panel` = n: records per n rows instead -- the order of a device that launches per row panel.)
        Returns (hidden [n_seq, H]: the residual stream of every sequence's last position, logits [n_seq, V]).  The head quantises its
        rows from the capture when head_captured (the batched int8 head), else with the oracle's own rounding (single-row GEMV head)."""
        cap = iter(captures) if captures is not None else None
        H, D, Hq, Hkv = self.H, self.D, self.Hq, self.Hkv
        rows, seg = [], []
        for s, p in zip(seq_ids, prompts):
            self.kc.setdefault(s, [[None, None] for _ in range(self.L)])
            start = 0 if self.kc[s][0][0] is None else self.kc[s][0][0].shape[1]
            seg.append((len(rows), len(p), start, s))
            rows.extend(p)
        S = len(rows)
        x = self.embed[np.asarray(rows, np.int64)].astype(F32)
        panel = panel or S
        panels = [(r0, min(panel, S - r0)) for r0 in range(0, S, panel)]
        for li, lw in enumerate(self.layers):
            xn = rms_norm(x, lw["ln1"], self.eps)
            qkv = np.empty((S, (Hq + 2 * Hkv) * D), F32)
            for r0, m in panels:
                q8, d8 = self._quant(np.ascontiguousarray(xn[r0:r0 + m]), cap, tie_tol)
                qkv[r0:r0 + m] = self._mm(q8, d8, (lw["q"], lw["k"], lw["v"]))
            attn = np.empty((S, Hq * D), F32)
            for row0, n, start, s in seg:
                blk = qkv[row0:row0 + n]
                qh = rms_norm(blk[:, :Hq * D].reshape(n, Hq, D), lw["qn"], self.eps)
                kh = rms_norm(blk[:, Hq * D:(Hq + Hkv) * D].reshape(n, Hkv, D), lw["kn"], self.eps)
                vh = blk[:, (Hq + Hkv) * D:].reshape(n, Hkv, D)
                c, sn = self.cos[start:start + n], self.sin[start:start + n]
                qh = rope_thd(qh, c, sn); kh = rope_thd(kh, c, sn)
                if self.kv_dtype == "f16":
                    kh, vh = f16_round(kh), f16_round(vh)
                kv = self.kc[s][li]
                kv[0] = kh.transpose(1, 0, 2) if kv[0] is None else np.concatenate([kv[0], kh.transpose(1, 0, 2)], axis=1)
                kv[1] = vh.transpose(1, 0, 2) if kv[1] is None else np.concatenate([kv[1], vh.transpose(1, 0, 2)], axis=1)
                T = kv[0].shape[1]
                qg = qh.transpose(1, 0, 2).reshape(Hkv, Hq // Hkv, n, D)
                sc = np.einsum("grnd,gld->grnl", qg, kv[0]).astype(F32) * F32(1.0 / math.sqrt(D))
                mask = np.arange(T)[None, :] > (start + np.arange(n))[:, None]            # key l visible to query i iff l <= start + i
                sc = np.where(mask[None, None], F32(-np.inf), sc)
                o = np.einsum("grnl,gld->grnd", softmax_last(sc), kv[1]).astype(F32)      # [Hkv, nrep, n, D]
                attn[row0:row0 + n] = o.reshape(Hq, n, D).transpose(1, 0, 2).reshape(n, Hq * D)
            for r0, m in panels:
                sl = slice(r0, r0 + m)
                # (attn_tol: the attention rows come from a matrix-core kernel with its own error budget, relative to the row: _quant_attn)
                if attn_tol is None: q8, d8 = self._quant(np.ascontiguousarray(attn[sl]), cap, tie_tol)
                else: q8, d8 = self._quant_attn(np.ascontiguousarray(attn[sl]), cap, attn_tol)
                x[sl] = (x[sl] + self._mm(q8, d8, (lw["o"],))).astype(F32)
                q8, d8 = self._quant(rms_norm(x[sl], lw["ln2"], self.eps), cap, tie_tol)
                gu = self._mm(q8, d8, (lw["gate"], lw["up"]))
                h = (silu(gu[:, :self.I]) * gu[:, self.I:]).astype(F32)
                q8, d8 = self._quant(h, cap, tie_tol)
                x[sl] = (x[sl] + self._mm(q8, d8, (lw["down"],))).astype(F32)
        hidden = np.stack([x[row0 + n - 1] for row0, n, _, _ in seg]).astype(F32)
        last = rms_norm(hidden, self.norm, self.eps)
        if self.head is None:
            logits = (last @ self.embed.T).astype(F32)
        else:
            q8, d8 = self._quant(last, cap if head_captured else None, tie_tol)
            logits = self._mm(q8, d8, (self.head,))
        assert cap is None or next(cap, None) is None, "the device quantised more activation rows than the pass has projections"
        return hidden, logits


def parse_captures(flat: np.ndarray):
    """cm_debug_read("q_capture") -> [(K, codes [M, K] int8, scales [M, K / 32] f32)]"""
    out, i = [], 0
    while i < flat.size:
        K, M = int(flat[i]), int(flat[i + 1]); i += 2
        codes = flat[i:i + M * K].astype(np.int8).reshape(M, K); i += M * K
        sc = flat[i:i + M * (K // 32)].astype(np.float32).reshape(M, K // 32); i += M * (K // 32)
        out.append((K, codes, sc))
    return out
