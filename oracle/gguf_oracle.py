"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the GGUF container and the ggml block formats the reference
loads through candle (`candle_core::quantized::{gguf_file, k_quants}` 0.11, a port of ggml's ggml-quants.c; the
crate is NOT under /root/reference, so the published ggml definitions are restated here).

Reference call sites: qwen3/model.rs:108-152 + qwen3/modeling.rs:242-282,593-606,678-696,821-960 (GGUF names,
metadata keys), hunyuan_dense/modeling.rs:14-95 (Gguf helper), ops/linear.rs:18-116 (QMatMul semantics: "dequantizes
weights to F32 and requires F32 input"; ISQ falls back to Q8_0 when K % 256 != 0).

Block formats (little endian):
  Q4_0  32 weights : f16 d; u8 qs[16] (element j: low nibble of qs[j], element j + 16: high nibble)   y = d * (q - 8)
  Q5_0  32 weights : f16 d; u32 qh; u8 qs[16] (5th bit of element j = bit j of qh)                    y = d * (q - 16)
  Q8_0  32 weights : f16 d; i8 qs[32]                                   y = d * q
  Q4_K  256 weights: f16 d; f16 dmin; u8 scales[12]; u8 qs[128]         y = d*sc_j*q - dmin*m_j   (8 sub-blocks of 32)
  Q6_K  256 weights: u8 ql[128]; u8 qh[64]; i8 scales[16]; f16 d        y = d*sc_j*(q - 32)        (16 sub-blocks of 16)
  Q3_K  256 weights: u8 hmask[32]; u8 qs[64]; u8 scales[12]; f16 d     y = d*(sc_j - 32)*(q2 - (hbit ? 0 : 4))   (16 sub-blocks of 16;
                     6-bit scales packed in 12 bytes, 2 low code bits in qs, the third -- inverted -- in hmask)
Dequantisers are exact restatements (dequantize_row_q4_0 / _q5_0 / _q8_0 / _q4_K / _q6_K / _q3_K).  quantize_q8_0 / _q4_0 / _q5_0 restate
quantize_row_q8_0_ref / _q4_0_ref / _q5_0_ref exactly; the Q4_K / Q6_K / Q3_K *quantisers* here are simple valid encoders for writing test files
(ggml's make_qkx2_quants search is not reproduced -- PARITY UNPINNED for K-quant ISQ, which crane_amd does not offer).
"""
import struct
from typing import Dict, List, Tuple

import numpy as np

GGML_F32, GGML_F16, GGML_Q8_0, GGML_Q4_K, GGML_Q6_K, GGML_BF16 = 0, 1, 8, 12, 14, 30
GGML_Q4_0, GGML_Q5_0 = 2, 6
GGML_Q3_K = 11
BLOCK = {GGML_F32: (1, 4), GGML_F16: (1, 2), GGML_BF16: (1, 2), GGML_Q8_0: (32, 34), GGML_Q4_K: (256, 144), GGML_Q6_K: (256, 210),
         GGML_Q4_0: (32, 18), GGML_Q5_0: (32, 22), GGML_Q3_K: (256, 110)}
TYPE_NAMES = {"f32": GGML_F32, "f16": GGML_F16, "bf16": GGML_BF16, "q8_0": GGML_Q8_0, "q4_k": GGML_Q4_K, "q6_k": GGML_Q6_K,
              "q4_0": GGML_Q4_0, "q5_0": GGML_Q5_0, "q3_k": GGML_Q3_K}


# ---------------------------------------------------------------------------------------------------------
# Q4_0 / Q5_0 (ggml-quants.c quantize_row_q4_0_ref / _q5_0_ref, dequantize_row_q4_0 / _q5_0)
# ---------------------------------------------------------------------------------------------------------
def _signed_max(xb: np.ndarray) -> np.ndarray:
    """the value (with its sign) of the FIRST element of each row that has the largest magnitude (`if (amax < fabsf(v))`)."""
    first = np.abs(xb).argmax(axis=1)
    return np.take_along_axis(xb, first[:, None], axis=1)[:, 0]


def _codes_q4_0(b: np.ndarray) -> np.ndarray:
    """[nb, 18] bytes -> int codes q - 8 in [-8, 7], element order of the row"""
    qs = b[:, 2:18]
    return np.concatenate([qs & 0xF, qs >> 4], axis=1).astype(np.int32) - 8


def _codes_q5_0(b: np.ndarray) -> np.ndarray:
    """[nb, 22] bytes -> int codes q - 16 in [-16, 15]"""
    qh = b[:, 2:6].copy().view(np.uint32)[:, 0]
    qs = b[:, 6:22]
    j = np.arange(16, dtype=np.uint32)
    h0 = ((qh[:, None] >> j) & 1).astype(np.int32) << 4                 # xh_0 = ((qh >> j) << 4) & 0x10
    h1 = ((qh[:, None] >> (j + 16)) & 1).astype(np.int32) << 4          # xh_1 = (qh >> (j + 12)) & 0x10
    lo = (qs & 0xF).astype(np.int32) | h0
    hi = (qs >> 4).astype(np.int32) | h1
    return np.concatenate([lo, hi], axis=1) - 16


def quantize_q4_0(x: np.ndarray) -> np.ndarray:
    """d = max / -8 (max = signed value of the largest magnitude), id = d ? 1/d : 0, q = min(15, (int8)(x * id + 8.5f))."""
    xb = np.asarray(x, np.float32).reshape(-1, 32)
    d = (_signed_max(xb) / np.float32(-8.0)).astype(np.float32)
    idv = np.where(d != 0, np.float32(1.0) / np.where(d != 0, d, 1), np.float32(0)).astype(np.float32)
    t = ((xb * idv[:, None]).astype(np.float32) + np.float32(8.5)).astype(np.float32)
    q = np.minimum(15, t.astype(np.int8).astype(np.int32)).astype(np.uint8)       # C cast: truncation toward zero
    out = np.zeros((xb.shape[0], 18), np.uint8)
    out[:, 0:2] = d.astype(np.float16).view(np.uint8).reshape(-1, 2)
    out[:, 2:] = q[:, :16] | (q[:, 16:] << 4)
    return out.reshape(-1)


def dequantize_q4_0(raw: np.ndarray, n: int) -> np.ndarray:
    b = np.asarray(raw, np.uint8).reshape(-1, 18)
    d = b[:, 0:2].copy().view(np.float16).astype(np.float32)
    return (_codes_q4_0(b).astype(np.float32) * d).astype(np.float32).reshape(-1)[:n]


def quantize_q5_0(x: np.ndarray) -> np.ndarray:
    """d = max / -16, q = min(31, (int8)(x * id + 16.5f)); low 4 bits in qs, the 5th bit of element j in bit j of qh."""
    xb = np.asarray(x, np.float32).reshape(-1, 32)
    d = (_signed_max(xb) / np.float32(-16.0)).astype(np.float32)
    idv = np.where(d != 0, np.float32(1.0) / np.where(d != 0, d, 1), np.float32(0)).astype(np.float32)
    t = ((xb * idv[:, None]).astype(np.float32) + np.float32(16.5)).astype(np.float32)
    q = np.minimum(31, t.astype(np.int8).astype(np.int32)).astype(np.uint32)
    out = np.zeros((xb.shape[0], 22), np.uint8)
    out[:, 0:2] = d.astype(np.float16).view(np.uint8).reshape(-1, 2)
    j = np.arange(16, dtype=np.uint32)
    qh = (((q[:, :16] >> 4) & 1) << j).sum(axis=1, dtype=np.uint32) | (((q[:, 16:] >> 4) & 1) << (j + 16)).sum(axis=1, dtype=np.uint32)
    out[:, 2:6] = qh.astype(np.uint32).view(np.uint8).reshape(-1, 4)
    out[:, 6:] = ((q[:, :16] & 0xF) | ((q[:, 16:] & 0xF) << 4)).astype(np.uint8)
    return out.reshape(-1)


def dequantize_q5_0(raw: np.ndarray, n: int) -> np.ndarray:
    b = np.asarray(raw, np.uint8).reshape(-1, 22)
    d = b[:, 0:2].copy().view(np.float16).astype(np.float32)
    return (_codes_q5_0(b).astype(np.float32) * d).astype(np.float32).reshape(-1)[:n]


# ---------------------------------------------------------------------------------------------------------
# Q8_0
# ---------------------------------------------------------------------------------------------------------
def quantize_q8_0(x: np.ndarray) -> np.ndarray:
    """quantize_row_q8_0_ref: d = amax / 127 (stored f16), q = roundf(x * id) with id = d ? 1/d : 0."""
    x = np.asarray(x, np.float32).reshape(-1, 32)
    amax = np.abs(x).max(axis=1)
    d = (amax / np.float32(127.0)).astype(np.float32)
    idv = np.where(d != 0, np.float32(1.0) / np.where(d != 0, d, 1), np.float32(0)).astype(np.float32)
    t = (x.reshape(-1) * np.repeat(idv, 32)).astype(np.float32)        # f32 product like the C code
    t64 = t.astype(np.float64)
    q = np.copysign(np.floor(np.abs(t64) + 0.5), t64).astype(np.int8)  # roundf: ties away from zero (exact in f64)
    out = np.zeros((x.shape[0], 34), np.uint8)
    out[:, 0:2] = d.astype(np.float16).view(np.uint8).reshape(-1, 2)
    out[:, 2:] = q.view(np.uint8).reshape(-1, 32)
    return out.reshape(-1)


def dequantize_q8_0(raw: np.ndarray, n: int) -> np.ndarray:
    b = np.asarray(raw, np.uint8).reshape(-1, 34)
    d = b[:, 0:2].copy().view(np.float16).astype(np.float32)              # [nb, 1]
    q = b[:, 2:].copy().view(np.int8).astype(np.float32)
    return (q * d).astype(np.float32).reshape(-1)[:n]


# ---------------------------------------------------------------------------------------------------------
# Q4_K
# ---------------------------------------------------------------------------------------------------------
def _get_scale_min_k4(j: int, s: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """get_scale_min_k4: 6-bit scales/mins packed in 12 bytes.  s: [nb, 12] uint8 -> (sc, m) [nb] each."""
    s = s.astype(np.uint16)
    if j < 4:
        return s[:, j] & 63, s[:, j + 4] & 63
    sc = (s[:, j + 4] & 0xF) | ((s[:, j - 4] >> 6) << 4)
    m = (s[:, j + 4] >> 4) | ((s[:, j] >> 6) << 4)
    return sc, m


def dequantize_q4_k(raw: np.ndarray, n: int) -> np.ndarray:
    b = np.asarray(raw, np.uint8).reshape(-1, 144)
    d = b[:, 0:2].copy().view(np.float16).astype(np.float32)[:, 0]
    dmin = b[:, 2:4].copy().view(np.float16).astype(np.float32)[:, 0]
    sc12 = b[:, 4:16]
    qs = b[:, 16:]
    y = np.zeros((b.shape[0], 256), np.float32)
    for p in range(4):                                       # 64 weights: low nibbles then high nibbles of 32 bytes
        q = qs[:, 32 * p:32 * p + 32]
        sc1, m1 = _get_scale_min_k4(2 * p, sc12)
        sc2, m2 = _get_scale_min_k4(2 * p + 1, sc12)
        d1 = (d * sc1.astype(np.float32)).astype(np.float32); mm1 = (dmin * m1.astype(np.float32)).astype(np.float32)
        d2 = (d * sc2.astype(np.float32)).astype(np.float32); mm2 = (dmin * m2.astype(np.float32)).astype(np.float32)
        y[:, 64 * p:64 * p + 32] = d1[:, None] * (q & 0xF).astype(np.float32) - mm1[:, None]
        y[:, 64 * p + 32:64 * p + 64] = d2[:, None] * (q >> 4).astype(np.float32) - mm2[:, None]
    return y.reshape(-1)[:n]


def quantize_q4_k(x: np.ndarray) -> np.ndarray:
    """A simple valid Q4_K encoder (min/max affine per sub-block, 6-bit scale/min against the super-block max)."""
    x = np.asarray(x, np.float32).reshape(-1, 8, 32)
    nb = x.shape[0]
    mn = np.minimum(x.min(axis=2), 0.0)                       # y = d*sc*q - dmin*m  with m >= 0
    mx = x.max(axis=2)
    scale = np.maximum(mx - mn, 1e-30) / 15.0                 # per sub-block
    mins = -mn
    d = (scale.max(axis=1) / 63.0).astype(np.float16).astype(np.float32)
    dmin = (mins.max(axis=1) / 63.0).astype(np.float16).astype(np.float32)
    sc = np.clip(np.rint(scale / np.where(d > 0, d, 1)[:, None]), 0, 63).astype(np.uint8)
    m = np.clip(np.rint(mins / np.where(dmin > 0, dmin, 1)[:, None]), 0, 63).astype(np.uint8)
    d1 = d[:, None] * sc
    m1 = dmin[:, None] * m
    q = np.clip(np.rint((x + m1[:, :, None]) / np.where(d1 > 0, d1, 1)[:, :, None]), 0, 15).astype(np.uint8)
    out = np.zeros((nb, 144), np.uint8)
    out[:, 0:2] = d.astype(np.float16).view(np.uint8).reshape(-1, 2)
    out[:, 2:4] = dmin.astype(np.float16).view(np.uint8).reshape(-1, 2)
    s = np.zeros((nb, 12), np.uint8)
    for j in range(8):
        if j < 4:
            s[:, j] |= sc[:, j] & 63
            s[:, j + 4] |= m[:, j] & 63
        else:
            s[:, j + 4] |= (sc[:, j] & 0xF) | ((m[:, j] & 0xF) << 4)
            s[:, j - 4] |= (sc[:, j] >> 4) << 6
            s[:, j] |= (m[:, j] >> 4) << 6
    out[:, 4:16] = s
    for p in range(4):
        out[:, 16 + 32 * p:16 + 32 * p + 32] = q[:, 2 * p] | (q[:, 2 * p + 1] << 4)
    return out.reshape(-1)


# ---------------------------------------------------------------------------------------------------------
# Q6_K
# ---------------------------------------------------------------------------------------------------------
def dequantize_q6_k(raw: np.ndarray, n: int) -> np.ndarray:
    b = np.asarray(raw, np.uint8).reshape(-1, 210)
    ql, qh = b[:, 0:128], b[:, 128:192]
    sc = b[:, 192:208].copy().view(np.int8).astype(np.float32)
    d = b[:, 208:210].copy().view(np.float16).astype(np.float32)[:, 0]
    y = np.zeros((b.shape[0], 256), np.float32)
    for h in range(2):                                       # two halves of 128 weights
        L, H, S = ql[:, 64 * h:64 * h + 64], qh[:, 32 * h:32 * h + 32], sc[:, 8 * h:8 * h + 8]
        l = np.arange(32)
        isx = l // 16
        q1 = ((L[:, l] & 0xF) | (((H[:, l] >> 0) & 3) << 4)).astype(np.int32) - 32
        q2 = ((L[:, l + 32] & 0xF) | (((H[:, l] >> 2) & 3) << 4)).astype(np.int32) - 32
        q3 = ((L[:, l] >> 4) | (((H[:, l] >> 4) & 3) << 4)).astype(np.int32) - 32
        q4 = ((L[:, l + 32] >> 4) | (((H[:, l] >> 6) & 3) << 4)).astype(np.int32) - 32
        base = 128 * h
        y[:, base + l] = (d[:, None] * S[:, isx + 0]) * q1
        y[:, base + 32 + l] = (d[:, None] * S[:, isx + 2]) * q2
        y[:, base + 64 + l] = (d[:, None] * S[:, isx + 4]) * q3
        y[:, base + 96 + l] = (d[:, None] * S[:, isx + 6]) * q4
    return y.astype(np.float32).reshape(-1)[:n]


def quantize_q6_k(x: np.ndarray) -> np.ndarray:
    """A simple valid Q6_K encoder (symmetric per 16-weight sub-block, int8 scale against the super-block)."""
    x = np.asarray(x, np.float32).reshape(-1, 16, 16)
    nb = x.shape[0]
    amax = np.abs(x).max(axis=2)                              # [nb, 16]
    sub = amax / 31.0
    d = (sub.max(axis=1) / 127.0).astype(np.float16).astype(np.float32)
    sc = np.clip(np.rint(sub / np.where(d > 0, d, 1)[:, None]), 0, 127).astype(np.int8)
    eff = d[:, None] * sc.astype(np.float32)
    q = np.clip(np.rint(x / np.where(eff > 0, eff, 1)[:, :, None]), -32, 31).astype(np.int32) + 32     # 0..63
    q = q.reshape(nb, 256).astype(np.uint8)
    out = np.zeros((nb, 210), np.uint8)
    for h in range(2):
        base = 128 * h
        l = np.arange(32)
        a1, a2, a3, a4 = q[:, base + l], q[:, base + 32 + l], q[:, base + 64 + l], q[:, base + 96 + l]
        out[:, 64 * h + l] = (a1 & 0xF) | ((a3 & 0xF) << 4)
        out[:, 64 * h + 32 + l] = (a2 & 0xF) | ((a4 & 0xF) << 4)
        out[:, 128 + 32 * h + l] = (a1 >> 4) | ((a2 >> 4) << 2) | ((a3 >> 4) << 4) | ((a4 >> 4) << 6)
    out[:, 192:208] = sc.view(np.uint8)
    out[:, 208:210] = d.astype(np.float16).view(np.uint8).reshape(-1, 2)
    return out.reshape(-1)


# ---------------------------------------------------------------------------------------------------------
# Q3_K (ggml-quants.c block_q3_K / dequantize_row_q3_K; candle k_quants.rs BlockQ3K)
# ---------------------------------------------------------------------------------------------------------
def _q3k_scales(sb: np.ndarray) -> np.ndarray:
    """[nb, 12] packed bytes -> [nb, 16] 6-bit scales (before the - 32), the aux[] shuffle of dequantize_row_q3_K."""
    a = np.ascontiguousarray(sb, np.uint8).view("<u4").astype(np.uint64)           # [nb, 3]
    k1, k2 = 0x03030303, 0x0F0F0F0F
    tmp = a[:, 2]
    o = np.zeros((a.shape[0], 4), np.uint64)
    o[:, 2] = ((a[:, 0] >> 4) & k2) | (((tmp >> 4) & k1) << 4)
    o[:, 3] = ((a[:, 1] >> 4) & k2) | (((tmp >> 6) & k1) << 4)
    o[:, 0] = (a[:, 0] & k2) | (((tmp >> 0) & k1) << 4)
    o[:, 1] = (a[:, 1] & k2) | (((tmp >> 2) & k1) << 4)
    return o.astype("<u4").view(np.uint8).reshape(-1, 16)


def _q3k_codes(b: np.ndarray) -> np.ndarray:
    """[nb, 110] raw Q3_K blocks -> [nb, 256] signed codes in [-4, 3], weight order."""
    hm, qs = b[:, 0:32], b[:, 32:96]
    y = np.zeros((b.shape[0], 256), np.int32)
    l = np.arange(32)
    for n in range(2):
        for j in range(4):
            lo = ((qs[:, 32 * n + l] >> (2 * j)) & 3).astype(np.int32)
            hb = (hm[:, l] >> (4 * n + j)) & 1
            y[:, 128 * n + 32 * j + l] = lo - np.where(hb != 0, 0, 4)
    return y


def dequantize_q3_k(raw: np.ndarray, n: int) -> np.ndarray:
    b = np.asarray(raw, np.uint8).reshape(-1, 110)
    sc = _q3k_scales(b[:, 96:108]).astype(np.int32) - 32                               # [nb, 16]
    d = b[:, 108:110].copy().view(np.float16).astype(np.float32)[:, 0]
    q = _q3k_codes(b).reshape(-1, 16, 16)
    dl = (d[:, None] * sc.astype(np.float32)).astype(np.float32)                       # d_all * (scales[is] - 32)
    return (dl[:, :, None] * q.astype(np.float32)).astype(np.float32).reshape(-1)[:n]


def quantize_q3_k(x: np.ndarray) -> np.ndarray:
    """A simple valid Q3_K encoder (symmetric per 16-weight sub-block, 6-bit scale against the super-block; negative scales are used
    when the sub-block's largest magnitude is positive, so that it lands on the code -4)."""
    x = np.asarray(x, np.float32).reshape(-1, 16, 16)
    nb = x.shape[0]
    i = np.abs(x).argmax(axis=2)
    mx = np.take_along_axis(x, i[:, :, None], axis=2)[:, :, 0]                         # signed value of the largest magnitude
    sub = -mx / 4.0                                                                      # that value -> code -4
    d = (np.abs(sub).max(axis=1) / 31.0).astype(np.float16).astype(np.float32)
    sc = np.clip(np.rint(sub / np.where(d > 0, d, 1)[:, None]), -32, 31).astype(np.int32)
    eff = d[:, None] * sc.astype(np.float32)
    q = np.clip(np.rint(x / np.where(eff != 0, eff, 1)[:, :, None]), -4, 3).astype(np.int32).reshape(nb, 256)
    u = (q + 4).astype(np.uint8)                                                         # 0..7: bit 2 = hmask bit, bits 0-1 = qs
    out = np.zeros((nb, 110), np.uint8)
    l = np.arange(32)
    for n in range(2):
        for j in range(4):
            v = u[:, 128 * n + 32 * j + l]
            out[:, 32 + 32 * n + l] |= (v & 3) << (2 * j)
            out[:, l] |= ((v >> 2) & 1) << (4 * n + j)
    s6 = (sc + 32).astype(np.uint8)                                                      # 0..63, packed like ggml: low 4 bits of scale j in byte
    for j in range(16):                                                                  # j (< 8: low nibble, >= 8: high nibble of byte j - 8),
        lo4, hi2 = s6[:, j] & 0xF, s6[:, j] >> 4                                         # high 2 bits in bytes 8..11
        if j < 8:
            out[:, 96 + j] |= lo4
        else:
            out[:, 96 + j - 8] |= lo4 << 4
        out[:, 96 + 8 + (j % 4)] |= hi2 << (2 * (j // 4))
    out[:, 108:110] = d.astype(np.float16).view(np.uint8).reshape(-1, 2)
    return out.reshape(-1)


def _bf16_bytes(x: np.ndarray) -> np.ndarray:
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    return r.view(np.uint8)


def quantize(x: np.ndarray, ggml_type: int) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float32).reshape(-1)
    if ggml_type == GGML_F32:
        return x.view(np.uint8)
    if ggml_type == GGML_F16:
        return x.astype(np.float16).view(np.uint8)
    if ggml_type == GGML_BF16:
        return _bf16_bytes(x)
    return {GGML_Q8_0: quantize_q8_0, GGML_Q4_K: quantize_q4_k, GGML_Q6_K: quantize_q6_k, GGML_Q4_0: quantize_q4_0,
            GGML_Q5_0: quantize_q5_0, GGML_Q3_K: quantize_q3_k}[ggml_type](x)


def dequantize(raw: np.ndarray, ggml_type: int, n: int) -> np.ndarray:
    raw = np.asarray(raw, np.uint8)
    if ggml_type == GGML_F32:
        return raw.view(np.float32)[:n].copy()
    if ggml_type == GGML_F16:
        return raw.view(np.float16)[:n].astype(np.float32)
    if ggml_type == GGML_BF16:
        return (raw.view(np.uint16)[:n].astype(np.uint32) << 16).view(np.float32)
    return {GGML_Q8_0: dequantize_q8_0, GGML_Q4_K: dequantize_q4_k, GGML_Q6_K: dequantize_q6_k, GGML_Q4_0: dequantize_q4_0,
            GGML_Q5_0: dequantize_q5_0, GGML_Q3_K: dequantize_q3_k}[ggml_type](raw, n)


# ---------------------------------------------------------------------------------------------------------
# GGUF v3 container
# ---------------------------------------------------------------------------------------------------------
GGUF_MAGIC = 0x46554747
(T_U8, T_I8, T_U16, T_I16, T_U32, T_I32, T_F32, T_BOOL, T_STR, T_ARR, T_U64, T_I64, T_F64) = range(13)
_SCALAR = {T_U8: "<B", T_I8: "<b", T_U16: "<H", T_I16: "<h", T_U32: "<I", T_I32: "<i", T_F32: "<f", T_BOOL: "<B",
           T_U64: "<Q", T_I64: "<q", T_F64: "<d"}


def _wstr(s: str) -> bytes:
    b = s.encode()
    return struct.pack("<Q", len(b)) + b


def _wval(v) -> bytes:
    """v = (type, value) or (T_ARR, (elem_type, [values]))."""
    t, x = v
    if t == T_STR:
        return struct.pack("<I", t) + _wstr(x)
    if t == T_ARR:
        et, items = x
        body = b"".join(_wstr(i) if et == T_STR else struct.pack(_SCALAR[et], i) for i in items)
        return struct.pack("<I", t) + struct.pack("<IQ", et, len(items)) + body
    return struct.pack("<I", t) + struct.pack(_SCALAR[t], x)


def write_gguf(path: str, metadata: Dict[str, tuple], tensors: List[Tuple[str, np.ndarray, int]], alignment: int = 32):
    """tensors: (name, f32 array [rows, cols] or [n], ggml_type).  GGUF dims are stored innermost-first."""
    infos, blobs, off = [], [], 0
    for name, arr, gt in tensors:
        arr = np.asarray(arr, np.float32)
        raw = quantize(arr, gt)
        dims = list(arr.shape)[::-1]
        infos.append((name, dims, gt, off))
        blobs.append(raw)
        off += (raw.size + alignment - 1) // alignment * alignment
    md = dict(metadata)
    md.setdefault("general.alignment", (T_U32, alignment))
    head = struct.pack("<IIQQ", GGUF_MAGIC, 3, len(infos), len(md))
    for k, v in md.items():
        head += _wstr(k) + _wval(v)
    for name, dims, gt, o in infos:
        head += _wstr(name) + struct.pack("<I", len(dims)) + b"".join(struct.pack("<Q", d) for d in dims) + struct.pack("<IQ", gt, o)
    pad = (-len(head)) % alignment
    with open(path, "wb") as f:
        f.write(head + b"\0" * pad)
        for raw in blobs:
            f.write(raw.tobytes())
            f.write(b"\0" * ((-raw.size) % alignment))


def read_gguf(path: str):
    """-> (metadata {key: python value}, tensors {name: (shape outermost-first, ggml_type, raw uint8)})."""
    data = np.fromfile(path, dtype=np.uint8)
    buf = data.tobytes()
    pos = 0

    def rd(fmt):
        nonlocal pos
        v = struct.unpack_from(fmt, buf, pos)
        pos += struct.calcsize(fmt)
        return v if len(v) > 1 else v[0]

    def rstr():
        nonlocal pos
        n = rd("<Q")
        s = buf[pos:pos + n].decode()
        pos += n
        return s

    def rval(t):
        if t == T_STR:
            return rstr()
        if t == T_ARR:
            et, n = rd("<IQ")
            return [rval(et) for _ in range(n)]
        return rd(_SCALAR[t])

    magic, version, n_t, n_kv = rd("<IIQQ")
    assert magic == GGUF_MAGIC and version in (2, 3), (hex(magic), version)
    md = {}
    for _ in range(n_kv):
        k = rstr()
        md[k] = rval(rd("<I"))
    infos = []
    for _ in range(n_t):
        name = rstr()
        nd = rd("<I")
        dims = [rd("<Q") for _ in range(nd)]
        gt, off = rd("<IQ")
        infos.append((name, dims[::-1], gt, off))
    align = md.get("general.alignment", 32)
    base = (pos + align - 1) // align * align
    out = {}
    for name, shape, gt, off in infos:
        n = int(np.prod(shape))
        be, bb = BLOCK[gt]
        nbytes = n // be * bb
        out[name] = (tuple(shape), gt, data[base + off:base + off + nbytes])
    return md, out


# ---------------------------------------------------------------------------------------------------------
# Qwen3 <-> GGUF naming (qwen3/modeling.rs:242-282, 593-606, 678-696, 821-960)
# ---------------------------------------------------------------------------------------------------------
def qwen3_gguf_names(cfg: dict) -> Dict[str, str]:
    """HF safetensors name -> GGUF tensor name."""
    m = {"model.embed_tokens.weight": "token_embd.weight", "model.norm.weight": "output_norm.weight"}
    if not cfg.get("tie_word_embeddings", True):
        m["lm_head.weight"] = "output.weight"
    for i in range(cfg["num_hidden_layers"]):
        p, g = f"model.layers.{i}.", f"blk.{i}."
        m.update({p + "self_attn.q_proj.weight": g + "attn_q.weight", p + "self_attn.k_proj.weight": g + "attn_k.weight",
                  p + "self_attn.v_proj.weight": g + "attn_v.weight", p + "self_attn.o_proj.weight": g + "attn_output.weight",
                  p + "self_attn.q_norm.weight": g + "attn_q_norm.weight", p + "self_attn.k_norm.weight": g + "attn_k_norm.weight",
                  p + "input_layernorm.weight": g + "attn_norm.weight", p + "post_attention_layernorm.weight": g + "ffn_norm.weight",
                  p + "mlp.gate_proj.weight": g + "ffn_gate.weight", p + "mlp.up_proj.weight": g + "ffn_up.weight",
                  p + "mlp.down_proj.weight": g + "ffn_down.weight"})
    return m


def qwen3_metadata(cfg: dict) -> Dict[str, tuple]:
    a = "qwen3"
    return {
        "general.architecture": (T_STR, a),
        f"{a}.block_count": (T_U32, cfg["num_hidden_layers"]),
        f"{a}.embedding_length": (T_U32, cfg["hidden_size"]),
        f"{a}.feed_forward_length": (T_U32, cfg["intermediate_size"]),
        f"{a}.attention.head_count": (T_U32, cfg["num_attention_heads"]),
        f"{a}.attention.head_count_kv": (T_U32, cfg["num_key_value_heads"]),
        f"{a}.attention.key_length": (T_U32, cfg.get("head_dim") or cfg["hidden_size"] // cfg["num_attention_heads"]),
        f"{a}.context_length": (T_U32, cfg.get("max_position_embeddings", 32768)),
        f"{a}.attention.layer_norm_rms_epsilon": (T_F32, cfg.get("rms_norm_eps", 1e-6)),
        f"{a}.rope.freq_base": (T_F32, cfg.get("rope_theta", 1e6)),
    }


def write_qwen3_gguf(path: str, cfg: dict, weights_f32: Dict[str, np.ndarray], type_of, want_qmats: bool = False):
    """type_of(gguf_name, shape) -> ggml type.  Returns the DEQUANTISED weights under their HF names: exactly what a
    loader of this file must compute with (the oracle forward runs on these).  With want_qmats also returns
    {hf name: QuantMatrix} for every quantised matrix (the vec_dot semantics)."""
    names = qwen3_gguf_names(cfg)
    tensors, deq, qm = [], {}, {}
    for hf, gg in names.items():
        w = np.asarray(weights_f32[hf], np.float32)
        gt = type_of(gg, w.shape)
        if w.ndim == 1:
            gt = GGML_F32
        tensors.append((gg, w, gt))
        raw = quantize(w, gt)
        deq[hf] = dequantize(raw, gt, w.size).reshape(w.shape)
        if want_qmats and w.ndim == 2 and gt in (GGML_Q8_0, GGML_Q4_K, GGML_Q6_K, GGML_Q4_0, GGML_Q5_0, GGML_Q3_K):
            qm[hf] = QuantMatrix(raw, gt, w.shape)
    write_gguf(path, qwen3_metadata(cfg), tensors)
    return (deq, qm) if want_qmats else deq


def qwen3_oracle_qmats(cfg: dict, qm: Dict[str, "QuantMatrix"]) -> dict:
    """{hf name: QuantMatrix} -> the (layer, name) -> [QuantMatrix...] map Qwen3Oracle.qmats expects."""
    out = {}
    for i in range(cfg["num_hidden_layers"]):
        p = f"model.layers.{i}."
        out[(i, "qkv")] = [qm[p + f"self_attn.{n}_proj.weight"] for n in ("q", "k", "v")]
        out[(i, "o")] = [qm[p + "self_attn.o_proj.weight"]]
        out[(i, "gate_up")] = [qm[p + "mlp.gate_proj.weight"], qm[p + "mlp.up_proj.weight"]]
        out[(i, "down")] = [qm[p + "mlp.down_proj.weight"]]
    head = "model.embed_tokens.weight" if cfg.get("tie_word_embeddings", True) else "lm_head.weight"
    if head in qm:
        out[(-1, "lm_head")] = [qm[head]]
    return out


# ---------------------------------------------------------------------------------------------------------
# Quantised mat-vec with QUANTISED ACTIVATIONS -- what candle's CPU QMatMul / ggml's vec_dot actually compute
# (k_quants.rs `vec_dot`; ggml-quants.c quantize_row_q8_0 / quantize_row_q8_K + ggml_vec_dot_q8_0_q8_0 /
# _q4_K_q8_K / _q6_K_q8_K): the activation row is quantised to the weight type's VecDotType (Q8_0 for Q8_0 weights,
# Q8_K for the K-quants), the products are summed as integers inside a block and scaled once per block.
# PARITY UNPINNED: the crate is not in /root/reference; q8_K's `iscale = -128 / max` follows candle 0.x k_quants.rs.
# ---------------------------------------------------------------------------------------------------------
def _roundf(t):
    t64 = np.asarray(t, np.float64)
    return np.copysign(np.floor(np.abs(t64) + 0.5), t64)


def quantize_act_q8_0(x: np.ndarray):
    """x [S, K] f32 -> (q int32 [S, K], d f32 [S, K/32] after the f16 round trip)."""
    x = np.asarray(x, np.float32)
    S, K = x.shape
    xb = x.reshape(S, K // 32, 32)
    amax = np.abs(xb).max(axis=2)
    d = (amax / np.float32(127.0)).astype(np.float32)
    idv = np.where(d != 0, np.float32(1.0) / np.where(d != 0, d, 1), np.float32(0)).astype(np.float32)
    q = _roundf((xb * idv[:, :, None]).astype(np.float32)).astype(np.int32).reshape(S, K)
    return q, d.astype(np.float16).astype(np.float32)


def quantize_act_q8_k(x: np.ndarray):
    """x [S, K] -> (q int32 [S, K], d f32 [S, K/256]); max = the signed value of the FIRST element with the largest |x|,
    iscale = -128 / max, q = min(127, nearest_int(iscale * x)) (round half to even), d = 1 / iscale."""
    x = np.asarray(x, np.float32)
    S, K = x.shape
    xb = x.reshape(S, K // 256, 256)
    ax = np.abs(xb)
    first = ax.argmax(axis=2)                                   # first occurrence of the maximum
    mx = np.take_along_axis(xb, first[:, :, None], axis=2)[:, :, 0]
    nz = mx != 0
    iscale = np.where(nz, np.float32(-128.0) / np.where(nz, mx, 1), np.float32(0)).astype(np.float32)
    t = (xb * iscale[:, :, None]).astype(np.float32)
    q = np.minimum(127, np.rint(t)).astype(np.int32)           # np.rint = half to even, like the magic-number nearest_int
    q = np.where(nz[:, :, None], q, 0).reshape(S, K)
    d = np.where(nz, np.float32(1.0) / np.where(nz, iscale, 1), np.float32(0)).astype(np.float32)
    return q, d


class QuantMatrix:
    """A [N, K] ggml-quantised matrix; `vecdot(x [S, K]) -> [S, N]` with ggml's quantised-activation semantics."""

    def __init__(self, raw: np.ndarray, ggml_type: int, shape):
        self.gt, self.shape = ggml_type, tuple(shape)
        N, K = self.shape
        if ggml_type in (GGML_Q4_0, GGML_Q5_0):
            # ggml_vec_dot_q4_0_q8_0 / _q5_0_q8_0: sumi = sum (q - 8 | 16) * q8 per block, sumf += sumi * d_w * d_x -- the Q8_0 x Q8_0
            # product of the integer codes q - offset (they fit an int8) with the block's own f16 scale
            bs = 18 if ggml_type == GGML_Q4_0 else 22
            b = np.asarray(raw, np.uint8).reshape(N * (K // 32), bs)
            self.d = b[:, 0:2].copy().view(np.float16).astype(np.float32)[:, 0].reshape(N, K // 32)
            self.q = (_codes_q4_0(b) if ggml_type == GGML_Q4_0 else _codes_q5_0(b)).reshape(N, K // 32, 32)
            self.gt = GGML_Q8_0
        elif ggml_type == GGML_Q8_0:
            b = np.asarray(raw, np.uint8).reshape(N, K // 32, 34)
            self.d = b[:, :, 0:2].copy().view(np.float16).astype(np.float32)[:, :, 0]
            self.q = b[:, :, 2:].copy().view(np.int8).astype(np.int32)                       # [N, nb, 32]
        elif ggml_type == GGML_Q4_K:
            b = np.asarray(raw, np.uint8).reshape(N * (K // 256), 144)
            self.d = b[:, 0:2].copy().view(np.float16).astype(np.float32)[:, 0].reshape(N, -1)
            self.dmin = b[:, 2:4].copy().view(np.float16).astype(np.float32)[:, 0].reshape(N, -1)
            sm = [_get_scale_min_k4(j, b[:, 4:16]) for j in range(8)]
            self.sc = np.stack([s for s, _ in sm], axis=1).astype(np.int32).reshape(N, -1, 8)
            self.mn = np.stack([m for _, m in sm], axis=1).astype(np.int32).reshape(N, -1, 8)
            qs = b[:, 16:].reshape(-1, 4, 32)
            q = np.stack([qs & 0xF, qs >> 4], axis=2)                                           # [nbt, 4, 2, 32]
            self.q = q.reshape(N, K // 256, 8, 32).astype(np.int32)
        elif ggml_type == GGML_Q6_K:
            deq_codes = _q6k_codes(np.asarray(raw, np.uint8).reshape(-1, 210))                  # [nbt, 256] in 0..63
            b = np.asarray(raw, np.uint8).reshape(-1, 210)
            self.q = (deq_codes.astype(np.int32) - 32).reshape(N, K // 256, 16, 16)
            self.sc = b[:, 192:208].copy().view(np.int8).astype(np.int32).reshape(N, -1, 16)
            self.d = b[:, 208:210].copy().view(np.float16).astype(np.float32)[:, 0].reshape(N, -1)
        elif ggml_type == GGML_Q3_K:
            # ggml_vec_dot_q3_K_q8_K: per super-block sum_j (scales[j] - 32) * sum_16 q q8 with q in [-4, 3], times d * d8 -- the Q6_K
            # arithmetic on narrower codes (16 sub-blocks of 16, int8 scales, one f16 d)
            b = np.asarray(raw, np.uint8).reshape(-1, 110)
            self.q = _q3k_codes(b).reshape(N, K // 256, 16, 16)
            self.sc = (_q3k_scales(b[:, 96:108]).astype(np.int32) - 32).reshape(N, -1, 16)
            self.d = b[:, 108:110].copy().view(np.float16).astype(np.float32)[:, 0].reshape(N, -1)
            self.gt = GGML_Q6_K
        else:
            raise ValueError("QuantMatrix: unsupported ggml type")

    def dequantize(self) -> np.ndarray:
        raise NotImplementedError

    def vecdot(self, x: np.ndarray) -> np.ndarray:
        x = np.asarray(x, np.float32)
        if x.ndim == 1:
            return self.vecdot(x[None])[0]
        N, K = self.shape
        S = x.shape[0]
        out = np.zeros((S, N), np.float32)
        if self.gt == GGML_Q8_0:
            xq, xd = quantize_act_q8_0(x)
            for s in range(S):
                isum = (self.q * xq[s].reshape(1, K // 32, 32)).sum(axis=2)                      # int, [N, nb]
                out[s] = (isum.astype(np.float32) * (self.d * xd[s][None, :]).astype(np.float32)).sum(axis=1, dtype=np.float32)
            return out
        xq, xd = quantize_act_q8_k(x)
        for s in range(S):
            q8 = xq[s].reshape(K // 256, 256)
            if self.gt == GGML_Q4_K:
                isum = (self.q * q8.reshape(1, -1, 8, 32)).sum(axis=3)                           # [N, nb, 8]
                sumi = (isum * self.sc).sum(axis=2)
                bs = q8.reshape(-1, 8, 32).sum(axis=2)                                           # bsums[2j] + bsums[2j+1]
                summs = (self.mn * bs[None]).sum(axis=2)
                dd = (xd[s][None, :] * self.d).astype(np.float32)
                dm = (xd[s][None, :] * self.dmin).astype(np.float32)
                out[s] = (dd * sumi.astype(np.float32) - dm * summs.astype(np.float32)).astype(np.float32).sum(axis=1, dtype=np.float32)
            else:
                isum = (self.q * q8.reshape(1, -1, 16, 16)).sum(axis=3)                          # [N, nb, 16]
                sumi = (isum * self.sc).sum(axis=2)
                dd = (xd[s][None, :] * self.d).astype(np.float32)
                out[s] = (dd * sumi.astype(np.float32)).astype(np.float32).sum(axis=1, dtype=np.float32)
        return out


def _q6k_codes(b: np.ndarray) -> np.ndarray:
    """[nb, 210] raw Q6_K blocks -> [nb, 256] unsigned 6-bit codes in weight order."""
    ql, qh = b[:, 0:128], b[:, 128:192]
    y = np.zeros((b.shape[0], 256), np.uint8)
    l = np.arange(32)
    for h in range(2):
        L, H = ql[:, 64 * h:64 * h + 64], qh[:, 32 * h:32 * h + 32]
        base = 128 * h
        y[:, base + l] = (L[:, l] & 0xF) | (((H[:, l] >> 0) & 3) << 4)
        y[:, base + 32 + l] = (L[:, l + 32] & 0xF) | (((H[:, l] >> 2) & 3) << 4)
        y[:, base + 64 + l] = (L[:, l] >> 4) | (((H[:, l] >> 4) & 3) << 4)
        y[:, base + 96 + l] = (L[:, l + 32] >> 4) | (((H[:, l] >> 6) & 3) << 4)
    return y


# ---------------------------------------------------------------------------------------------------------
# Qwen 3.5 family <-> llama.cpp `qwen35` GGUF layout (qwen3_5/model.rs:155-325, modeling.rs:375-411,684-775):
#   block / final / per-head q,k norms stored with the +1 already folded; ssm_norm plain
#   linear-attention blocks: attn_qkv (in_proj_qkv), attn_gate (z), ssm_beta (b), ssm_alpha (a), ssm_conv1d [conv_dim, k],
#   ssm_a = -exp(A_log), ssm_dt.bias, ssm_norm, ssm_out; full-attention: attn_q ([q | gate] per head), attn_k/v/output
#   value-head axis is CHUNKED (index = replica * num_k_heads + key_head), HF is Interleaved (key_head * vpg + replica)
# ---------------------------------------------------------------------------------------------------------
def _chunk_perm(NK: int, NV: int) -> np.ndarray:
    """perm[c] = HF (interleaved) value-head index stored at chunked position c."""
    vpg = NV // NK
    return np.array([(c % NK) * vpg + (c // NK) for c in range(NV)], dtype=np.int64)


def qwen35_metadata(cfg: dict) -> Dict[str, tuple]:
    a, t = "qwen35", cfg.get("text_config", cfg)
    rp = t.get("rope_parameters", {})
    D = t["head_dim"]
    md = {
        "general.architecture": (T_STR, a),
        f"{a}.block_count": (T_U32, t["num_hidden_layers"]), f"{a}.embedding_length": (T_U32, t["hidden_size"]),
        f"{a}.feed_forward_length": (T_U32, t["intermediate_size"]), f"{a}.attention.head_count": (T_U32, t["num_attention_heads"]),
        f"{a}.attention.head_count_kv": (T_U32, t["num_key_value_heads"]), f"{a}.attention.key_length": (T_U32, D),
        f"{a}.context_length": (T_U32, t.get("max_position_embeddings", 262144)),
        f"{a}.attention.layer_norm_rms_epsilon": (T_F32, t.get("rms_norm_eps", 1e-6)),
        f"{a}.rope.freq_base": (T_F32, rp.get("rope_theta", 1e7)),
        f"{a}.rope.dimension_count": (T_U32, int(D * rp.get("partial_rotary_factor", 0.25))),
        f"{a}.rope.dimension_sections": (T_ARR, (T_I32, list(rp.get("mrope_section", [11, 11, 10])) + [0])),
        f"{a}.full_attention_interval": (T_U32, t.get("full_attention_interval", 4)),
        f"{a}.ssm.conv_kernel": (T_U32, t.get("linear_conv_kernel_dim", 4)), f"{a}.ssm.state_size": (T_U32, t["linear_key_head_dim"]),
        f"{a}.ssm.group_count": (T_U32, t["linear_num_key_heads"]), f"{a}.ssm.time_step_rank": (T_U32, t["linear_num_value_heads"]),
        f"{a}.ssm.inner_size": (T_U32, t["linear_num_value_heads"] * t["linear_value_head_dim"]),
    }
    return md


def write_qwen35_gguf(path: str, cfg: dict, w: Dict[str, np.ndarray], type_of) -> Dict[str, np.ndarray]:
    """HF-named f32 weights -> qwen35 GGUF.  Returns the HF-named, HF-ordered weights a loader of the file computes with
    (matrices dequantised from the file and un-permuted; A_log = log(-ssm_a))."""
    t = cfg.get("text_config", cfg)
    L, NK, NV = t["num_hidden_layers"], t["linear_num_key_heads"], t["linear_num_value_heads"]
    K, V, interval = t["linear_key_head_dim"], t["linear_value_head_dim"], t.get("full_attention_interval", 4)
    KD = NK * K
    perm = _chunk_perm(NK, NV)
    inv = np.argsort(perm)
    vrow = (perm[:, None] * V + np.arange(V)[None, :]).reshape(-1)          # chunked position -> HF row of the value axis
    vinv = (inv[:, None] * V + np.arange(V)[None, :]).reshape(-1)
    tensors, out = [], dict(w)

    def put(gg, arr, hf=None, undo=None, force=None):
        arr = np.asarray(arr, np.float32)
        gt = force if force is not None else (GGML_F32 if arr.ndim == 1 else type_of(gg, arr.shape))
        tensors.append((gg, arr, gt))
        if hf is not None:
            d = dequantize(quantize(arr, gt), gt, arr.size).reshape(arr.shape)
            out[hf] = undo(d) if undo else d

    put("token_embd.weight", w["model.embed_tokens.weight"], "model.embed_tokens.weight")
    put("output_norm.weight", w["model.norm.weight"] + 1.0)
    if "lm_head.weight" in w and not t.get("tie_word_embeddings", cfg.get("tie_word_embeddings", False)):
        put("output.weight", w["lm_head.weight"], "lm_head.weight")
    for i in range(L):
        p, g = f"model.layers.{i}.", f"blk.{i}."
        put(g + "attn_norm.weight", w[p + "input_layernorm.weight"] + 1.0)
        put(g + "post_attention_norm.weight", w[p + "post_attention_layernorm.weight"] + 1.0)
        for n in ("gate", "up", "down"):
            put(g + f"ffn_{n}.weight", w[p + f"mlp.{n}_proj.weight"], p + f"mlp.{n}_proj.weight")
        if (i + 1) % interval == 0:
            a = p + "self_attn."
            put(g + "attn_q.weight", w[a + "q_proj.weight"], a + "q_proj.weight")
            put(g + "attn_k.weight", w[a + "k_proj.weight"], a + "k_proj.weight")
            put(g + "attn_v.weight", w[a + "v_proj.weight"], a + "v_proj.weight")
            put(g + "attn_output.weight", w[a + "o_proj.weight"], a + "o_proj.weight")
            put(g + "attn_q_norm.weight", w[a + "q_norm.weight"] + 1.0)
            put(g + "attn_k_norm.weight", w[a + "k_norm.weight"] + 1.0)
        else:
            a = p + "linear_attn."
            qkv = w[a + "in_proj_qkv.weight"]
            rows = np.concatenate([np.arange(2 * KD), 2 * KD + vrow])
            rinv = np.concatenate([np.arange(2 * KD), 2 * KD + vinv])
            put(g + "attn_qkv.weight", qkv[rows], a + "in_proj_qkv.weight", undo=lambda d, r=rinv: d[r])
            put(g + "attn_gate.weight", w[a + "in_proj_z.weight"][vrow], a + "in_proj_z.weight", undo=lambda d: d[vinv])
            put(g + "ssm_beta.weight", w[a + "in_proj_b.weight"][perm], force=GGML_F32)
            put(g + "ssm_alpha.weight", w[a + "in_proj_a.weight"][perm], force=GGML_F32)
            conv = w[a + "conv1d.weight"].reshape(2 * KD + NV * V, -1)
            put(g + "ssm_conv1d.weight", conv[rows], force=GGML_F32)
            ssm_a = (-np.exp(w[a + "A_log"].astype(np.float32))).astype(np.float32)
            put(g + "ssm_a", ssm_a[perm], force=GGML_F32)
            out[a + "A_log"] = np.log(-ssm_a).astype(np.float32)
            put(g + "ssm_dt.bias", w[a + "dt_bias"][perm], force=GGML_F32)
            put(g + "ssm_norm.weight", w[a + "norm.weight"], force=GGML_F32)
            put(g + "ssm_out.weight", w[a + "out_proj.weight"][:, vrow], a + "out_proj.weight", undo=lambda d: d[:, vinv])
    write_gguf(path, qwen35_metadata(cfg), tensors)
    return out
