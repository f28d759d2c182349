"""CPU: oracle/gguf_oracle.py -- ggml block formats (hand-built known answers + quantise/dequantise bounds) and the
GGUF v3 container round trip."""
import numpy as np
import pytest

from oracle import gguf_oracle as G


def test_q8_0_known_block():
    x = np.zeros(32, np.float32); x[0] = 127.0; x[1] = -63.5; x[2] = 0.49; x[3] = 2.5
    raw = G.quantize_q8_0(x)
    assert raw.size == 34 and raw[:2].view(np.float16)[0] == np.float16(1.0)
    q = raw[2:].view(np.int8)
    assert q[0] == 127 and q[1] == -64 and q[2] == 0 and q[3] == 3          # roundf: halves away from zero
    np.testing.assert_array_equal(G.dequantize_q8_0(raw, 32)[:4], [127, -64, 0, 3])
    assert G.dequantize_q8_0(G.quantize_q8_0(np.zeros(32)), 32).tolist() == [0.0] * 32


def test_q4_0_and_q5_0_known_blocks():
    """Hand-built blocks of the two 32-weight legacy formats (ggml-quants.c block_q4_0 / block_q5_0): nibble order, the 5th-bit
    word, the -8 / -16 offsets; and the reference quantisers' rule (d = signed max / -8 | -16, truncating cast, clamp)."""
    b = np.zeros(18, np.uint8)
    b[0:2] = np.array([0.5], np.float16).view(np.uint8)
    b[2] = 0x3A                                               # element 0: q = 0xA -> (10 - 8) * 0.5; element 16: q = 3 -> (3 - 8) * 0.5
    b[2 + 15] = 0xF0                                          # element 15: q = 0 -> -4.0; element 31: q = 15 -> 3.5
    y = G.dequantize_q4_0(b, 32)
    assert y[0] == 1.0 and y[16] == -2.5 and y[15] == -4.0 and y[31] == 3.5 and y[1] == -4.0
    c = np.zeros(22, np.uint8)
    c[0:2] = np.array([2.0], np.float16).view(np.uint8)
    c[2:6] = np.array([(1 << 3) | (1 << (16 + 5))], np.uint32).view(np.uint8)     # 5th bits of elements 3 and 21
    c[6 + 3] = 0x07                                           # element 3: q = 7 | 16 = 23 -> (23 - 16) * 2; element 19: q = 0 -> -32
    c[6 + 5] = 0x90                                           # element 5: q = 0 -> -32; element 21: q = 9 | 16 = 25 -> 18
    z = G.dequantize_q5_0(c, 32)
    assert z[3] == 14.0 and z[19] == -32.0 and z[5] == -32.0 and z[21] == 18.0
    x = np.zeros(32, np.float32); x[4] = -8.0; x[7] = 7.0; x[9] = 0.49; x[30] = 3.5
    raw = G.quantize_q4_0(x)                                  # max = -8 -> d = 1: q = (int8)(x + 8.5) clamped to 15
    assert raw[:2].view(np.float16)[0] == np.float16(1.0)
    np.testing.assert_array_equal(G.dequantize_q4_0(raw, 32)[[4, 7, 9, 30, 0]], [-8.0, 7.0, 0.0, 4.0, 0.0])
    x[4] = -16.0; x[7] = 15.9
    raw = G.quantize_q5_0(x)                                  # d = 1: 15.9 + 16.5 = 32.4 -> 32 -> clamped to 31 -> 15
    np.testing.assert_array_equal(G.dequantize_q5_0(raw, 32)[[4, 7, 9, 30]], [-16.0, 15.0, 0.0, 4.0])
    rng = np.random.default_rng(5)
    w = (rng.standard_normal((3, 64)) * 0.1).astype(np.float32)
    for t, tol in ((G.GGML_Q4_0, 0.08), (G.GGML_Q5_0, 0.04)):
        raw = G.quantize(w, t)
        deq = G.dequantize(raw, t, w.size).reshape(w.shape)
        assert np.abs(deq - w).max() <= tol * np.abs(w).max() + 1e-7
        xv = rng.standard_normal(64).astype(np.float32)
        qm = G.QuantMatrix(raw, t, w.shape)                   # Q4_0 x Q8_0 / Q5_0 x Q8_0 integer dots
        assert np.abs(qm.vecdot(xv) - deq @ xv).max() < 0.02 * np.abs(deq @ xv).max() + 1e-3


def test_q3_k_known_block():
    """A hand-built Q3_K super-block (ggml-quants.c block_q3_K: hmask[32], qs[64], scales[12], d): the inverted third bit, the 2-bit
    planes, the 6-bit scale packing (low nibbles in bytes 0-7, high pairs in bytes 8-11) and the - 32; then the encoder / QuantMatrix."""
    b = np.zeros(110, np.uint8)
    b[108:110] = np.array([0.5], np.float16).view(np.uint8)
    hm, qs, sc = b[0:32], b[32:96], b[96:108]
    # scales: sub-block 0 = 35 (-> +3), sub-block 1 = 0 (-> -32), sub-block 9 = 63 (-> +31), sub-block 15 = 33 (-> +1)
    sc[0] = 35 & 0xF; sc[8] |= (35 >> 4) << 0                 # j = 0: low nibble of byte 0, high bits = bits 0-1 of byte 8
    sc[1] = 0
    sc[1] |= (63 & 0xF) << 4; sc[8 + 1] |= (63 >> 4) << 4     # j = 9: high nibble of byte 1, high bits = bits 4-5 of byte 9
    sc[7] |= (33 & 0xF) << 4; sc[8 + 3] |= (33 >> 4) << 6     # j = 15: high nibble of byte 7, high bits = bits 6-7 of byte 11
    # element 0 (n 0, j 0, l 0): low bits 3, hmask bit 0 set -> code 3; element 1: low 2, bit clear -> 2 - 4 = -2
    qs[0] = 3; hm[0] |= 1
    qs[1] = 2
    # element 16 (sub-block 1; n 0, j 0, l 16): low 1, bit clear -> -3 under the scale -32
    qs[16] = 1
    # element 144 (sub-block 9: n 1, j 0, l 16): qs byte 32 + 16, shift 0; hmask bit 4 of byte 16 -> code 1
    qs[32 + 16] |= 1; hm[16] |= 1 << 4
    # element 255 (sub-block 15: n 1, j 3, l 31): qs byte 32 + 31 shift 6; hmask bit 7 of byte 31: low 0, bit clear -> -4
    y = G.dequantize_q3_k(b, 256)
    assert y[0] == 0.5 * 3 * 3 and y[1] == 0.5 * 3 * -2
    assert y[16] == 0.5 * -32 * -3
    assert y[144] == 0.5 * 31 * 1
    assert y[255] == 0.5 * 1 * -4
    assert y[2] == 0.5 * 3 * -4                               # an all-zero element decodes to -4 under its sub-block's scale
    rng = np.random.default_rng(7)
    w = (rng.standard_normal((3, 512)) * 0.1).astype(np.float32)
    raw = G.quantize(w, G.GGML_Q3_K)
    assert raw.size == 3 * 2 * 110
    deq = G.dequantize(raw, G.GGML_Q3_K, w.size).reshape(w.shape)
    assert np.abs(deq - w).max() <= 0.3 * np.abs(w).max()     # three bits
    xv = rng.standard_normal(512).astype(np.float32)
    qm = G.QuantMatrix(raw, G.GGML_Q3_K, w.shape)             # Q3_K x Q8_K integer dots (the Q6_K arithmetic on narrower codes)
    assert np.abs(qm.vecdot(xv) - deq @ xv).max() < 0.02 * np.abs(deq @ xv).max() + 1e-3


def test_q4_k_known_block():
    b = np.zeros(144, np.uint8)
    b[0:2] = np.array([2.0], np.float16).view(np.uint8)       # d
    b[2:4] = np.array([0.5], np.float16).view(np.uint8)       # dmin
    s = b[4:16]
    s[0] = 3; s[4] = 2                                        # sub-block 0: sc 3, m 2
    s[1] = 1 | (1 << 6); s[5] = 4 | (2 << 6)                  # sub-block 1: sc 1, m 4; high bits feed sub-blocks 4/5
    s[8] = (5 & 0xF) | ((7 & 0xF) << 4)                       # sub-block 4: sc = 5 | (s[0]>>6)<<4 = 5, m = 7 | (s[4]>>6)<<4 = 7
    s[9] = (6 & 0xF) | ((1 & 0xF) << 4)                       # sub-block 5: sc = 6 | (1<<4) = 22, m = 1 | (2<<4) = 33
    qs = b[16:]
    qs[0] = 0x21                                              # weight 0: q 1 (sub-block 0); weight 32: q 2 (sub-block 1)
    qs[64 + 3] = 0xF4                                         # weight 128+3: q 4 (sub-block 4); weight 160+3: q 15 (sub-block 5)
    y = G.dequantize_q4_k(b, 256)
    assert y[0] == 2.0 * 3 * 1 - 0.5 * 2 and y[1] == -1.0
    assert y[32] == 2.0 * 1 * 2 - 0.5 * 4 and y[33] == -2.0
    assert y[128 + 3] == 2.0 * 5 * 4 - 0.5 * 7 and y[128 + 4] == -3.5
    assert y[160 + 3] == 2.0 * 22 * 15 - 0.5 * 33


def test_q6_k_known_block():
    b = np.zeros(210, np.uint8)
    b[208:210] = np.array([0.5], np.float16).view(np.uint8)
    sc = b[192:208].view(np.int8)
    sc[:] = np.arange(1, 17)
    sc[9] = -3
    ql, qh = b[0:128], b[128:192]
    ql[5] = 0x9C; qh[5] = 0b10_01_11_00                       # l=5: q1 = 0xC|0<<4, q3 = 0x9|(1<<4); q2/q4 take ql[37]
    ql[37] = 0x07                                             # q2 = 7 | (3<<4) = 55, q4 = 0 | (2<<4) = 32
    ql[64 + 20] = 0x01; qh[32 + 20] = 0b00_00_00_10           # second half, l=20 (is=1): q1 = 1 | (2<<4) = 33
    y = G.dequantize_q6_k(b, 256)
    assert y[5] == 0.5 * 1 * (12 - 32)                        # scale index 0
    assert y[32 + 5] == 0.5 * 3 * (55 - 32)                   # scale index 2
    assert y[64 + 5] == 0.5 * 5 * (25 - 32)                   # scale index 4
    assert y[96 + 5] == 0.5 * 7 * (32 - 32)
    assert y[128 + 20] == 0.5 * (-3) * (33 - 32)              # second half: scales 8.., is = 1 -> sc[9]
    assert y[0] == 0.5 * 1 * (0 - 32)


@pytest.mark.parametrize("name,tol", [("q8_0", 0.005), ("q6_k", 0.03), ("q4_k", 0.09), ("f16", 1e-3), ("bf16", 4e-3), ("f32", 0.0)])
def test_quantise_dequantise_bound(name, tol):
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(4096) * 0.05).astype(np.float32)
    gt = G.TYPE_NAMES[name]
    raw = G.quantize(x, gt)
    assert raw.size == x.size // G.BLOCK[gt][0] * G.BLOCK[gt][1]
    y = G.dequantize(raw, gt, x.size)
    assert np.abs(y - x).max() <= tol * np.abs(x).max() + 1e-12
    np.testing.assert_array_equal(G.dequantize(G.quantize(y, gt), gt, x.size), y) if name in ("q8_0", "f16", "bf16", "f32") else None


def test_gguf_container_round_trip(tmp_path):
    rng = np.random.default_rng(2)
    a = rng.standard_normal((8, 256)).astype(np.float32)
    b = rng.standard_normal(64).astype(np.float32)
    md = {"general.architecture": (G.T_STR, "qwen3"), "qwen3.block_count": (G.T_U32, 7), "qwen3.rope.freq_base": (G.T_F32, 1e6),
          "tokenizer.ggml.tokens": (G.T_ARR, (G.T_STR, ["a", "bc"])), "x.arr": (G.T_ARR, (G.T_I32, [1, -2, 3]))}
    p = str(tmp_path / "t.gguf")
    G.write_gguf(p, md, [("blk.0.attn_q.weight", a, G.GGML_Q4_K), ("output_norm.weight", b, G.GGML_F32), ("w8", a, G.GGML_Q8_0)])
    m2, t2 = G.read_gguf(p)
    assert m2["general.architecture"] == "qwen3" and m2["qwen3.block_count"] == 7 and m2["tokenizer.ggml.tokens"] == ["a", "bc"]
    assert m2["x.arr"] == [1, -2, 3] and abs(m2["qwen3.rope.freq_base"] - 1e6) < 1
    shape, gt, raw = t2["blk.0.attn_q.weight"]
    assert shape == (8, 256) and gt == G.GGML_Q4_K
    np.testing.assert_array_equal(raw, G.quantize(a, G.GGML_Q4_K))
    np.testing.assert_array_equal(G.dequantize(t2["output_norm.weight"][2], G.GGML_F32, 64), b)
    assert t2["w8"][0] == (8, 256) and t2["w8"][2].size == 8 * 8 * 34


def test_c_restatement_of_the_quantisers_and_vec_dot_equals_the_numpy_one():
    """oracle/c/q8_ref.c (the fast checker of the int8-MFMA decode-group tests, oracle/qgroup_oracle.py) against this file's numpy
    restatement, which carries the known-answer blocks: codes and f16-rounded scales bit for bit, ggml_vec_dot_q8_0_q8_0 to the f32
    summation order (numpy sums the blocks pairwise, the C code in ascending order like ggml)."""
    import ctypes as C
    import os
    from oracle import c_oracle
    if not os.path.exists(c_oracle.SO):
        pytest.skip("oracle/c not built")
    from oracle.qgroup_oracle import QMat, _p
    lib = c_oracle._lib()
    rng = np.random.default_rng(0)
    w = (rng.standard_normal((96, 512)) * np.abs(rng.standard_normal((96, 512)))).astype(np.float32)
    w[3, 32:64] = 0                                   # an all-zero block: d = 0, id = 0
    w[5, 0] = -w[5, 1:32].max() * 2                   # signed maximum negative (Q4_0 / Q5_0: d > 0)
    x = (rng.standard_normal((7, 512)) * 3).astype(np.float32)
    for name, fmt in (("q8_0", 8), ("q4_0", 2), ("q5_0", 6)):
        gt = G.TYPE_NAMES[name]
        qm = QMat(lib, w, fmt)
        ref = G.QuantMatrix(G.quantize(w, gt), gt, w.shape)
        assert np.array_equal(qm.q.reshape(96, 16, 32), ref.q) and np.array_equal(qm.d, ref.d), name
        xq, xd = G.quantize_act_q8_0(x)
        xq8, xd = np.ascontiguousarray(xq.astype(np.int8)), np.ascontiguousarray(xd)
        out = np.empty((7, 96), np.float32)
        assert lib.qc_vec_dot_q8_rows(_p(qm.q, C.c_int8), _p(qm.d, C.c_float), 96, 512, _p(xq8, C.c_int8), _p(xd, C.c_float), 7, _p(out, C.c_float)) == 0
        want = ref.vecdot(x)
        assert np.abs(out - want).max() <= 1e-6 * np.abs(want).max(), name
    # the synthetic-weight generator as f32 (crane_amd/synth.py, bit-identical)
    from crane_amd import synth
    got = np.empty((5, 64), np.float32)
    lib.qc_synth_f32(b"model.layers.1.mlp.down_proj.weight", 3, 0.02, 0.0, 5, 64, _p(got, C.c_float))
    want = synth.bf16_bits_to_f32(synth.synth_bf16_bits("model.layers.1.mlp.down_proj.weight", 320, 3, 0.02, 0.0)).reshape(5, 64)
    assert np.array_equal(got, want)
