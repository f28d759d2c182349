"""TEST INFRASTRUCTURE ONLY -- per-token symmetric int8 / int4 KV quantisation, restating
crane-core/src/models/qwen3_5/kv_cache.rs:253-301 (quantize_per_token, dequantize_per_token, pack/unpack_nibbles):
  qmax = 2^(bits-1) - 1 (127 | 7), offset = 2^(bits-1) (128 | 8)
  scale = amax(|x|, last dim) * (1/qmax) + 1e-8              (candle affine: f32 multiply then f32 add)
  code  = round(x / scale) + offset                           (f32::round = half away from zero; codes in [1, 2*qmax+1])
  int4 : byte = lo + 16 * hi for adjacent (even, odd) pairs
  deq   = (code - offset) * scale
The attention of the step that appends a token already sees the dequantised value (QuantKvCache::append returns
dequantize(full cache), :318-322)."""
import numpy as np

F32 = np.float32


def quantize_per_token(x: np.ndarray, bits: int):
    """x [..., D] f32 -> (codes uint8 [..., D] (int8) or [..., D/2] (int4, packed), scale f32 [..., 1])."""
    assert bits in (4, 8)
    qmax, offset = F32((1 << (bits - 1)) - 1), F32(1 << (bits - 1))
    x = np.asarray(x, F32)
    amax = np.abs(x).max(axis=-1, keepdims=True).astype(F32)
    scale = (amax * F32(1.0 / float(qmax)) + F32(1e-8)).astype(F32)
    t = (x / scale).astype(F32)
    r = np.copysign(np.floor(np.abs(t.astype(np.float64)) + 0.5), t).astype(F32)       # f32::round
    codes = (r + offset).astype(F32)
    if bits == 8:
        return codes.astype(np.uint8), scale
    lo, hi = codes[..., 0::2], codes[..., 1::2]
    return (lo + hi * F32(16.0)).astype(np.uint8), scale


def dequantize_per_token(codes: np.ndarray, scale: np.ndarray, bits: int) -> np.ndarray:
    offset = F32(1 << (bits - 1))
    if bits == 8:
        q = codes.astype(F32)
    else:
        byte = codes.astype(F32)
        hi = np.floor(byte * F32(1.0 / 16.0))
        lo = byte - hi * F32(16.0)
        q = np.stack([lo, hi], axis=-1).reshape(*codes.shape[:-1], codes.shape[-1] * 2)
    return ((q - offset) * scale).astype(F32)


def roundtrip(x: np.ndarray, bits: int) -> np.ndarray:
    c, s = quantize_per_token(x, bits)
    return dequantize_per_token(c, s, bits)
