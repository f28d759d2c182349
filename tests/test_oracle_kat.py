"""Pin the CPU oracle with the reference's own known-answer tests.

Each test cites the reference test it replays (paths under /root/reference/crane-core/src).
The reference sources are NOT read at run time; the expected values are the ones written
in those tests (hand-computed formulas / deterministic arange inputs).
"""
import math

import numpy as np

from oracle.qwen3_oracle import (Qwen3Config, Qwen3Oracle, apply_repeat_penalty, flash_gqa_attention,
                                 naive_gqa_attention, rms_norm, rope_thd, rotary_tables, silu)

F32 = np.float32


def test_inv_freq_values():
    """models/modules/rotary.rs:166-189: dim=8, theta=1e4 => inv_freq=[1,.1,.01,.001]; pos 1."""
    cos, sin = rotary_tables(8, 2, 10000.0)
    for i, f in enumerate([1.0, 0.1, 0.01, 0.001]):
        assert abs(cos[1, i] - math.cos(F32(f))) < 1e-5
        assert abs(sin[1, i] - math.sin(F32(f))) < 1e-5


def test_inv_freq_monotonic_decay():
    """rotary.rs:191-209."""
    _, sin = rotary_tables(64, 4, 10000.0)
    assert np.all(np.diff(sin[1]) < 0)


def test_table_values_at_specific_positions():
    """rotary.rs:211-235: dim=4, theta=100 => inv_freq=[1.0, 0.1]; positions 0,1,5,10."""
    cos, sin = rotary_tables(4, 16, 100.0)
    for pos in (0, 1, 5, 10):
        for i, f in enumerate([pos * 1.0, pos * 0.1]):
            assert abs(cos[pos, i] - math.cos(F32(f))) < 1e-5
            assert abs(sin[pos, i] - math.sin(F32(f))) < 1e-5


def test_apply_rotation_formula_manual():
    """rotary.rs:372-409: q=[1,2,3,4] at position 3, contiguous half-split pairing."""
    cos, sin = rotary_tables(4, 8, 100.0)
    q = np.array([1, 2, 3, 4], dtype=F32).reshape(1, 1, 4)       # [S, H, D]
    out = rope_thd(q, cos[3:4], sin[3:4])[0, 0]
    c0, s0, c1, s1 = math.cos(3.0), math.sin(3.0), math.cos(0.3), math.sin(0.3)
    exp = [1 * c0 - 3 * s0, 2 * c1 - 4 * s1, 1 * s0 + 3 * c0, 2 * s1 + 4 * c1]
    assert np.allclose(out, exp, atol=1e-5)


def test_rope_position_zero_is_identity():
    """rotary.rs:352-369."""
    cos, sin = rotary_tables(8, 4, 10000.0)
    x = np.arange(2 * 3 * 8, dtype=F32).reshape(1, 6, 8)
    assert np.abs(rope_thd(x, cos[:1], sin[:1]) - x).max() < 1e-5


def test_rope_preserves_norm():
    cos, sin = rotary_tables(16, 32, 1e6)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((5, 3, 16)).astype(F32)
    y = rope_thd(x, cos[7:12], sin[7:12])
    assert np.allclose(np.linalg.norm(x, axis=-1), np.linalg.norm(y, axis=-1), rtol=1e-5)


def _arange_qkv(num_heads, kv_heads, sq, kv_len, hd):
    q = (np.arange(num_heads * sq * hd, dtype=F32) * F32(0.037)).reshape(num_heads, sq, hd)
    k = (np.arange(kv_heads * kv_len * hd, dtype=F32) * F32(0.021)).reshape(kv_heads, kv_len, hd)
    v = (np.arange(kv_heads * kv_len * hd, dtype=F32) * F32(0.013)).reshape(kv_heads, kv_len, hd)
    return q, k, v


def test_flash_attn_decode_matches_naive_gqa_softmax():
    """qwen3/modeling.rs:1467-1526: kv_heads 2, n_rep 2, head_dim 4, kv_len 5, arange inputs, tol 1e-4."""
    q, k, v = _arange_qkv(4, 2, 1, 5, 4)
    scale = 1.0 / math.sqrt(4)
    assert np.abs(naive_gqa_attention(q, k, v, scale) - flash_gqa_attention(q, k, v, scale)).max() < 1e-4


def test_flash_attn_prefill_matches_naive_sdpa():
    """qwen3/modeling.rs:1633-1720: seq_len 3, kv_offset in {0, 2}, causal, tol 1e-4."""
    scale = 1.0 / math.sqrt(4)
    for kv_offset in (0, 2):
        q, k, v = _arange_qkv(4, 2, 3, kv_offset + 3, 4)
        a = naive_gqa_attention(q, k, v, scale, causal_offset=kv_offset)
        b = flash_gqa_attention(q, k, v, scale, causal_offset=kv_offset)
        assert np.abs(a - b).max() < 1e-4


def _tiny_model(seed=0):
    """tiny_config of qwen3/modeling.rs:1386-1403 (vocab 32, hidden 16, 1 layer, 4q/2kv heads, head_dim 4)."""
    cfg = Qwen3Config(vocab_size=32, hidden_size=16, intermediate_size=32, num_hidden_layers=1,
                      num_attention_heads=4, num_key_value_heads=2, head_dim=4,
                      max_position_embeddings=64, rms_norm_eps=1e-6, rope_theta=10000.0,
                      tie_word_embeddings=False)
    rng = np.random.default_rng(seed)
    w = {"model.embed_tokens.weight": rng.standard_normal((32, 16)).astype(F32),
         "model.norm.weight": np.ones(16, F32),
         "lm_head.weight": (rng.standard_normal((32, 16)) / 4).astype(F32)}
    p = "model.layers.0."
    w[p + "self_attn.q_proj.weight"] = (rng.standard_normal((16, 16)) / 4).astype(F32)
    w[p + "self_attn.k_proj.weight"] = (rng.standard_normal((8, 16)) / 4).astype(F32)
    w[p + "self_attn.v_proj.weight"] = (rng.standard_normal((8, 16)) / 4).astype(F32)
    w[p + "self_attn.o_proj.weight"] = (rng.standard_normal((16, 16)) / 4).astype(F32)
    w[p + "self_attn.q_norm.weight"] = np.ones(4, F32)
    w[p + "self_attn.k_norm.weight"] = np.ones(4, F32)
    w[p + "mlp.gate_proj.weight"] = (rng.standard_normal((32, 16)) / 4).astype(F32)
    w[p + "mlp.up_proj.weight"] = (rng.standard_normal((32, 16)) / 4).astype(F32)
    w[p + "mlp.down_proj.weight"] = (rng.standard_normal((16, 32)) / 6).astype(F32)
    w[p + "input_layernorm.weight"] = np.ones(16, F32)
    w[p + "post_attention_layernorm.weight"] = np.ones(16, F32)
    return cfg, w


def test_chunked_prefill_matches_single():
    """qwen3/modeling.rs:1763-1801: prefill [1..5] then decode 6 == prefill [1,2,3]+[4,5] then decode 6; tol 1e-4."""
    cfg, w = _tiny_model()
    a, b = Qwen3Oracle(cfg, w), Qwen3Oracle(cfg, w)
    a.forward([1, 2, 3, 4, 5], 0)
    out_single = a.forward([6], 5)
    b.forward([1, 2, 3], 0)
    b.forward([4, 5], 3)
    out_chunked = b.forward([6], 5)
    assert out_single.shape == out_chunked.shape == (32,)
    assert np.abs(out_single - out_chunked).max() < 1e-4


def test_forward_is_deterministic_and_last_position_only():
    """qwen3/modeling.rs:1737-1760 (determinism) + :1032-1035 (logits of the last position only)."""
    cfg, w = _tiny_model(1)
    m = Qwen3Oracle(cfg, w)
    x = m.forward([1, 2, 3], 0)
    m.clear_kv_cache()
    y = m.forward([1, 2, 3], 0)
    assert np.array_equal(x, y) and x.shape == (cfg.vocab_size,)


def test_generate_feeds_whole_prompt_then_single_tokens():
    """qwen3/model.rs:299-304,348: step 0 = whole prompt at start_pos 0; returns prompt ++ generated."""
    cfg, w = _tiny_model(2)
    m = Qwen3Oracle(cfg, w)
    out = m.generate([3, 1, 4], 4)
    assert out[:3] == [3, 1, 4] and len(out) == 7
    # replay by hand
    m2 = Qwen3Oracle(cfg, w)
    toks = [3, 1, 4]
    toks.append(int(m2.forward(toks, 0).argmax()))
    for _ in range(3):
        toks.append(int(m2.forward(toks[-1:], len(toks) - 1).argmax()))
    assert toks == out


def test_rms_norm_formula():
    """crane-core/tests/qwen3_5_norms.rs:63-110: deterministic values(), x/sqrt(mean(x^2)+eps)*w."""
    def values(n, seed):
        return np.array([(((i * 37 + seed * 11) % 97) / 97.0 - 0.5) * 4.0 for i in range(n)], dtype=F32)
    x, w = values(64, 1).reshape(4, 16), values(16, 2) + F32(1)
    got = rms_norm(x, w, 1e-6)
    for r in range(4):
        ms = float(np.mean(x[r].astype(np.float64) ** 2))
        exp = x[r] / math.sqrt(ms + 1e-6) * w
        assert np.abs(got[r] - exp).max() < 1e-5


def test_silu_and_repeat_penalty_arithmetic():
    """kernels/cuda/fused_ops.cu:47-49 and crane-serve engine/sampling.rs:501-560 (penalty arithmetic)."""
    x = np.array([-2.0, 0.0, 3.0], dtype=F32)
    assert np.allclose(silu(x), x / (1 + np.exp(-x)), atol=1e-7)
    l = np.array([2.0, -2.0, 1.0, 0.5], dtype=F32)
    out = apply_repeat_penalty(l, 2.0, [0, 1, 0])
    assert np.allclose(out, [1.0, -4.0, 1.0, 0.5])          # positive /p, negative *p, once per distinct token


def test_kv_quant_per_token_rules():
    """qwen3_5/kv_cache.rs:253-301: codes in [1, 2*qmax+1], scale = amax/qmax + 1e-8, nibble packing lo + 16*hi."""
    from oracle import kv_quant_oracle as Q
    x = np.array([[1.0, -2.0, 0.5, 4.0, -4.0, 0.0, 3.99, -0.01]], np.float32)
    c8, s8 = Q.quantize_per_token(x, 8)
    assert s8.shape == (1, 1) and s8[0, 0] == np.float32(4.0) * np.float32(1.0 / 127.0) + np.float32(1e-8)
    assert c8.dtype == np.uint8 and c8.min() >= 1 and c8[0, 3] == 255 and c8[0, 4] == 1 and c8[0, 5] == 128
    c4, s4 = Q.quantize_per_token(x, 4)
    assert c4.shape == (1, 4) and s4[0, 0] == np.float32(4.0) * np.float32(1.0 / 7.0) + np.float32(1e-8)
    codes = np.round(x / s4) + 8
    assert c4[0, 0] == int(codes[0, 0]) + 16 * int(codes[0, 1]) and c4[0, 1] == int(codes[0, 2]) + 16 * 15
    for bits, tol in ((8, 0.5 / 127 + 1e-6), (4, 0.5 / 7 + 1e-6)):
        y = Q.roundtrip(x, bits)
        assert np.abs(y - x).max() <= tol * 4.0
    z = np.zeros((2, 8), np.float32)
    assert np.all(Q.roundtrip(z, 8) == 0) and np.all(Q.roundtrip(z, 4) == 0)       # scale = 1e-8, codes = offset
