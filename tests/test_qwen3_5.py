"""Qwen 3.5 hybrid (Gated Delta Net + gated softmax attention): oracle KATs / HF golden (CPU) and
HIP-path parity through the C ABI (GPU)."""
import os

import numpy as np
import pytest

from crane_amd import configs, synth
from oracle import qwen3_5_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
F32 = np.float32


def rel(a, ref):
    return float(np.abs(a - ref).max() / np.abs(ref).max())


def _load():
    g = np.load(os.path.join(GOLD, "qwen3_5_tiny-qwen3.5.npz"))
    cfg = configs.get_config("tiny-qwen3.5")
    return g, cfg, synth.synth_weights_f32(cfg, seed=int(g["seed"][0]))


# ---------------------------------------------------------------- CPU: oracle pinned ----------
def test_derived_dims_of_qwen38_27b():
    """qwen3_5/config.rs:332-364: q_proj rows 12288, conv_dim 10240, value_dim 6144, key_dim 2048, rot_dim 64,
    16 full layers at ratio 3:1."""
    c = O.Qwen35Config.from_json(configs.get_config("qwen3.8-27b"))
    assert c.num_attention_heads * c.head_dim * 2 == 12288
    assert (c.conv_dim, c.value_dim, c.key_dim, c.rot_dim) == (10240, 6144, 2048, 64)
    full = [i for i in range(c.num_hidden_layers) if c.layer_is_full(i)]
    assert len(full) == 16 and full[0] == 3 and all(b - a == 4 for a, b in zip(full, full[1:]))


def test_unit_offset_and_gated_norm_formulas():
    """crane-core/tests/qwen3_5_norms.rs:72-231: block norm = x/rms*(1+w); L2-norm = x/sqrt(sum+eps);
    gated norm = rms_norm(y, w) * silu(z) with a PLAIN weight."""
    def values(n, seed):
        return np.array([(((i * 37 + seed * 11) % 97) / 97.0 - 0.5) * 4.0 for i in range(n)], dtype=F32)
    x, w = values(48, 1).reshape(3, 16), values(16, 2) * F32(0.1)
    got = O.rms_norm_1p(x, w, 1e-6)
    for r in range(3):
        ms = float(np.mean(x[r].astype(np.float64) ** 2))
        assert np.abs(got[r] - x[r] / np.sqrt(ms + 1e-6) * (1 + w)).max() < 1e-5
    n = O.l2_norm(x)
    assert np.allclose(np.sum(n * n, axis=-1), np.sum(x * x, -1) / (np.sum(x * x, -1) + 1e-6), atol=1e-5)
    z = values(48, 3).reshape(3, 16)
    gn = O.rms_norm_plain(x, w + 1, 1e-6) * O.silu(z)
    assert np.allclose(gn, (x / np.sqrt(np.mean(x * x, -1, keepdims=True) + 1e-6)) * (w + 1) * (z / (1 + np.exp(-z))), atol=1e-5)


def test_conv_chunked_equals_decode_steps():
    """ops/gdn/conv.rs:135-328 (chunk / decode equivalence, tol 1e-6) exercised through the layer: a prompt fed
    as one chunk, as two chunks, or token by token must give the same logits (also the GDN state hand-over)."""
    g, cfg, w = _load()
    c = O.Qwen35Config.from_json(cfg)
    ids = g["prompt"].tolist()
    a = O.Qwen35Oracle(c, w).forward(ids, 0)
    o2 = O.Qwen35Oracle(c, w)
    o2.forward(ids[:7], 0)
    b = o2.forward(ids[7:], 7)
    o3 = O.Qwen35Oracle(c, w)
    for i, t in enumerate(ids):
        d = o3.forward([t], i)
    assert rel(b, a) < 1e-5 and rel(d, a) < 1e-5


def test_interleaved_head_pairing():
    """ops/gdn/layer.rs:280-326: with 2 key heads and v_per_group 2, value heads [0,1] pair with key head 0 and
    [2,3] with key head 1 (HF 'Interleaved' order = repeat_interleave)."""
    k = np.array([[[10.0], [20.0]]], dtype=F32)            # [S=1, NK=2, K=1]
    assert np.repeat(k, 2, axis=1)[0, :, 0].tolist() == [10.0, 10.0, 20.0, 20.0]


def test_oracle_matches_hf_golden():
    g, cfg, w = _load()
    o = O.Qwen35Oracle(O.Qwen35Config.from_json(cfg), w)
    ids = g["prompt"].tolist()
    assert rel(o.forward(ids, 0), g["prefill_logits"]) < 2e-5
    assert rel(o.forward(g["decode_token"].tolist(), len(ids)), g["decode_logits"]) < 2e-5
    assert o.generate(ids, len(g["greedy_tokens"]) - len(ids)) == g["greedy_tokens"].tolist()


def test_oracle_matches_hf_golden_at_27b_geometry():
    """The oracle against HF Qwen3_5ForCausalLM at the REAL Qwen3.8-27B layer geometry (qwen3_5/config.rs:298-324: n_rep 6,
    3 value heads per key head, head_dim 256 with 64 rotary dims, K = 5120 / 17408), 4 layers and a 4096-entry vocabulary --
    the configuration the HIP path is compared with the oracle on in test_hip_qwen38_27b_geometry.  (~1 min: 1.56 G
    synthetic parameters.)"""
    g = np.load(os.path.join(GOLD, "qwen3_5_qwen3.8-27b-geom4.npz"))
    cfg = dict(configs.get_config("qwen3.8-27b"), num_hidden_layers=4, vocab_size=4096, max_position_embeddings=4096)
    w = synth.synth_weights_f32(cfg, seed=int(g["seed"][0]))
    o = O.Qwen35Oracle(O.Qwen35Config.from_json(cfg), w)
    ids = g["prompt"].tolist()
    assert rel(o.forward(ids, 0), g["prefill_logits"]) < 2e-5
    assert rel(o.forward(g["decode_token"].tolist(), len(ids)), g["decode_logits"]) < 2e-5
    assert o.generate(ids, len(g["greedy_tokens"]) - len(ids)) == g["greedy_tokens"].tolist()


# ---------------------------------------------------------------- GPU: HIP path ----------------
# HIP path vs the HF f32 forward: f32 pages 1e-4; f16 pages (the default, the benchmarked mode) the north-star bar 1e-3;
# bf16 pages (opt-in) one bf16 epsilon -- not claimed to meet the bar
HF_TOL = {"f32": 1e-4, "f16": 1e-3, "bf16": 4e-3}


@pytest.mark.gpu
@pytest.mark.parametrize("kv", ["f32", "f16", "bf16"])
def test_hip_qwen35_matches_hf_golden_and_oracle(kv):
    from crane_amd.backend import GenerationConfig, Model
    g, cfg, w = _load()
    m = Model.synthetic(cfg, seed=0, max_seq_len=256, max_seqs=3, kv_dtype=kv)
    try:
        ids = g["prompt"].tolist()
        tol = HF_TOL[kv]
        assert rel(m.forward_step(ids, 0)[0, 0], g["prefill_logits"]) < tol
        assert rel(m.forward_step(g["decode_token"].tolist(), len(ids))[0, 0], g["decode_logits"]) < tol
        n_new = len(g["greedy_tokens"]) - len(ids)
        assert m.generate(ids, GenerationConfig.greedy(n_new)) == g["greedy_tokens"].tolist()
        assert m.generate(ids, GenerationConfig.greedy(n_new), sync_every=4) == g["greedy_tokens"].tolist()
        # longer run against the oracle, crossing a KV page (64) on the full-attention layers
        o = O.Qwen35Oracle(O.Qwen35Config.from_json(cfg), w, kv_dtype=kv)
        long_ids = configs.synthetic_prompt(80, cfg["vocab_size"])
        m.clear_kv_cache()
        assert rel(m.forward_step(long_ids, 0)[0, 0], o.forward(long_ids, 0)) < (1e-4 if kv == "f32" else 5e-4)
    finally:
        m.close()


@pytest.mark.gpu
def test_hip_qwen35_sequences_and_state():
    """per-sequence GDN state: fork copies it, clear resets it, truncation to a non-zero length is refused."""
    from crane_amd._lib import CraneError
    from crane_amd.backend import Model
    g, cfg, w = _load()
    o = O.Qwen35Oracle(O.Qwen35Config.from_json(cfg), w)
    m = Model.synthetic(cfg, seed=0, max_seq_len=256, max_seqs=4, kv_dtype="f32")
    try:
        ids = g["prompt"].tolist()
        s1 = m.seq_alloc()
        m.seq_forward(s1, ids, 0)
        s2 = m.seq_fork(s1)
        l1, _ = m.seq_forward(s1, [11], len(ids))
        l2, _ = m.seq_forward(s2, [23], len(ids))
        o.forward(ids, 0); r1 = o.forward([11], len(ids))
        o.forward(ids, 0); r2 = o.forward([23], len(ids))
        assert rel(l1, r1) < 1e-4 and rel(l2, r2) < 1e-4
        with pytest.raises(CraneError):
            m.seq_truncate(s1, 5)
        m.seq_truncate(s1, 0)
        l3, _ = m.seq_forward(s1, ids, 0)                     # state really was reset
        assert rel(l3, o.forward(ids, 0)) < 1e-4
    finally:
        m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 19, 64, 130])
def test_hip_qwen35_prefill_equals_token_serial(n):
    """MFMA prefill (in_proj GEMM + sequential delta-rule scan + D=256 gated flash attention) == token-serial
    GEMV path == oracle; even/odd lengths exercise the conv-window parity split."""
    from crane_amd.backend import Model
    g, cfg, w = _load()
    o = O.Qwen35Oracle(O.Qwen35Config.from_json(cfg), w)
    m = Model.synthetic(cfg, seed=0, max_seq_len=256, max_seqs=2, kv_dtype="f32", prefill_chunk=96)
    try:
        ids = configs.synthetic_prompt(n, cfg["vocab_size"])
        a = m.forward_step(ids, 0)[0, 0]
        nxt = m.forward_step([5], n)[0, 0]
        m.debug_set("no_prefill", 1)
        try:
            m.clear_kv_cache()
            b = m.forward_step(ids, 0)[0, 0]
        finally:
            m.debug_set("no_prefill", 0)
        ref = o.forward(ids, 0)
        assert rel(a, b) < 1e-4 and rel(a, ref) < 1e-4
        assert rel(nxt, o.forward([5], n)) < 1e-4
    finally:
        m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,n,chunk", [("tiny-qwen3.5", 64, 0), ("tiny-qwen3.5", 65, 0), ("tiny-qwen3.5", 201, 0), ("tiny-qwen3.5", 300, 128),
                                          ("qwen3.8-27b", 193, 0), ("qwen3.5-0.8b", 1024, 0)])
def test_chunk_parallel_delta_rule_scan_equals_the_sequential_scan(name, n, chunk):
    """Prompts of >= 64 tokens run the Gated-Delta-Net recurrence in its chunk-parallel (WY / UT-transform) form on the f32 matrix
    cores (gdn_chunk_prep_kernel + gdn_chunk_scan_kernel; the reference names it as the prompt-path algorithm,
    ops/gdn/backend.rs:100-104, HF = torch_chunk_gated_delta_rule): same logits as the sequential scan of the same handle
    (cm_debug_set("gdn_chunked", 0)) to f32 summation order, the SAME recurrent state afterwards (a decode step on top), and the
    oracle's token-serial recurrence (ops/gdn/backend.rs:90-156) where the CPU side is cheap.  Whole chunks, a one-token tail
    chunk, ragged tails, prompts split over several passes (state carried between them), the 27B head geometry (3 value heads
    per key head, 48 heads) and a 1024-token prompt at the Qwen3.5-0.8B widths."""
    from crane_amd.backend import Model
    cfg = dict(configs.get_config(name))
    if cfg.get("num_hidden_layers", 0) > 4:
        cfg.update(num_hidden_layers=4, vocab_size=4096, max_position_embeddings=4096)
        if "layer_types" in cfg: cfg["layer_types"] = cfg["layer_types"][:4]
    kw = dict(prefill_chunk=chunk) if chunk else {}
    m = Model.synthetic(cfg, seed=0, max_seq_len=max(256, n + 64), max_seqs=2, kv_dtype="f32", **kw)
    try:
        V = cfg["vocab_size"]
        ids = configs.synthetic_prompt(n, V)
        outs = []
        for mode in (1, 0):
            m.debug_set("gdn_chunked", mode)
            m.clear_kv_cache()
            a = m.forward_step(ids, 0)[0, 0].copy()
            b = m.forward_step([5], n)[0, 0].copy()              # reads the recurrent state the prompt pass left
            outs.append((a, b))
        assert rel(outs[0][0], outs[1][0]) < 1e-4, rel(outs[0][0], outs[1][0])
        assert rel(outs[0][1], outs[1][1]) < 1e-4, rel(outs[0][1], outs[1][1])
        assert int(outs[0][0].argmax()) == int(outs[1][0].argmax())
        if name == "tiny-qwen3.5":
            w = synth.synth_weights_f32(cfg, seed=0)
            o = O.Qwen35Oracle(O.Qwen35Config.from_json(cfg), w)
            assert rel(outs[0][0], o.forward(ids, 0)) < 1e-4
            assert rel(outs[0][1], o.forward([5], n)) < 1e-4
        else:
            # real widths (4 layers of Qwen3.5-0.8B x 1024 tokens; the 27B head geometry -- 48 value heads, 3 per key head -- x 193):
            # the chunk-parallel scan against the token-serial recurrence of the C oracle (oracle/c/qwen35_cpu.c, ops/gdn/backend.rs:90-156;
            # pinned on the HF fixtures by tests/test_c_oracle.py), prompt logits and the decode step on the state the prompt left
            from oracle import c_oracle
            co = c_oracle.CQwen35(cfg, seed=0, max_seq=n + 64)
            try:
                ra = co.forward(ids, 0)
                rb = co.forward([5], n)
            finally:
                co.close()
            assert rel(outs[0][0], ra) < 1e-4, rel(outs[0][0], ra)
            assert rel(outs[0][1], rb) < 1e-4, rel(outs[0][1], rb)
            assert int(outs[0][0].argmax()) == int(ra.argmax())
    finally:
        m.close()


@pytest.mark.gpu
def test_hip_qwen35_batched_decode():
    """batched step on the hybrid model: per-sequence GDN state slots + per-sequence KV pages in one pass."""
    from crane_amd.backend import Model
    g, cfg, w = _load()
    V = cfg["vocab_size"]
    m = Model.synthetic(cfg, seed=0, max_seq_len=256, max_seqs=6, kv_dtype="f32")
    try:
        oracles, seqs, lens = [], [], []
        for i in range(5):
            n = 4 + 9 * i
            ids = [(13 * i + 7 * k + 3) % V for k in range(n)]
            o = O.Qwen35Oracle(O.Qwen35Config.from_json(cfg), w)
            o.forward(ids, 0)
            s = m.seq_alloc(); m.seq_forward(s, ids, 0, want_logits=False)
            oracles.append(o); seqs.append(s); lens.append(n)
        toks = [(5 + i) % V for i in range(5)]
        for step in range(3):
            lg, greedy = m.step_batch_decode(seqs, toks)
            nxt = []
            for i in range(5):
                ref = oracles[i].forward([toks[i]], lens[i] + step)
                assert rel(lg[i, 0], ref) < 1e-4, (i, step)
                nxt.append(int(ref.argmax()))
            toks = nxt
    finally:
        m.close()


@pytest.mark.gpu
def test_hip_qwen35_batched_decode_gemm_path():
    """24 hybrid sequences in one batched step: from 17 sequences on the projections (in_proj / out_proj of the GDN layers,
    QKV / o_proj of the gated-attention layers, the MLP) run as MFMA GEMMs over the batch rows.  Every row against its own
    oracle, and against a twin set of sequences stepped through the batched-GEMV path."""
    from crane_amd.backend import Model
    g, cfg, w = _load()
    V = cfg["vocab_size"]
    nseq = 24
    m = Model.synthetic(cfg, seed=0, max_seq_len=256, max_seqs=2 * nseq + 2, kv_dtype="f32")
    try:
        oracles, sa, sb, lens = [], [], [], []
        for i in range(nseq):
            n = 2 + (5 * i) % 37
            ids = [(13 * i + 7 * k + 3) % V for k in range(n)]
            if i % 6 == 0:                                  # (the numpy oracle is slow: every sixth row is checked against it)
                o = O.Qwen35Oracle(O.Qwen35Config.from_json(cfg), w)
                o.forward(ids, 0)
                oracles.append((i, o))
            for dst in (sa, sb):
                s_ = m.seq_alloc(); m.seq_forward(s_, ids, 0, want_logits=False); dst.append(s_)
            lens.append(n)
        toks = [(5 + 3 * i) % V for i in range(nseq)]
        for step in range(2):
            m.debug_set("batch_gemm_min", 0)
            lg_v, gr_v = m.step_batch_decode(sa, toks)
            m.debug_set("batch_gemm_min", 9)
            lg_m, gr_m = m.step_batch_decode(sb, toks)
            assert not np.array_equal(lg_m, lg_v)
            for i in range(nseq):
                assert rel(lg_m[i, 0], lg_v[i, 0]) < 1e-4, (step, i)
            for i, o in oracles:
                assert rel(lg_m[i, 0], o.forward([toks[i]], lens[i] + step)) < 1e-4, (step, i)
            assert [int(t) for t in gr_m] == [int(t) for t in gr_v]
            toks = [int(t) for t in gr_m]
    finally:
        m.close()


@pytest.mark.gpu
def test_hip_qwen38_27b_geometry():
    """Real Qwen3.8-27B layer geometry (qwen3_5/config.rs:298-324: H 5120, 24q/4kv x 256, 16 key / 48 value GDN heads,
    I 17408) with 4 layers and a small vocabulary so the CPU oracle stays fast -- the analogue of the reference's
    crane-core/tests/qwen3_5_gqa_grouped_decode.rs (algebra at 27B geometry).  Exercises n_rep = 6, 3 value heads per
    key head, K = 5120 / 17408 GEMV chunk counts, decode + MFMA prefill + batched decode."""
    from crane_amd.backend import Model
    cfg = dict(configs.get_config("qwen3.8-27b"), num_hidden_layers=4, vocab_size=4096, max_position_embeddings=4096)
    w = synth.synth_weights_f32(cfg, seed=0)
    o = O.Qwen35Oracle(O.Qwen35Config.from_json(cfg), w)
    m = Model.synthetic(cfg, seed=0, max_seq_len=256, max_seqs=3, kv_dtype="f32")
    try:
        ids = configs.synthetic_prompt(37, cfg["vocab_size"])
        assert rel(m.forward_step(ids, 0)[0, 0], o.forward(ids, 0)) < 1e-4            # prefill path
        for pos, t in enumerate([5, 9, 2], start=37):                                   # decode path
            assert rel(m.forward_step([t], pos)[0, 0], o.forward([t], pos)) < 1e-4
        o2 = O.Qwen35Oracle(O.Qwen35Config.from_json(cfg), w)
        s1, s2 = m.seq_alloc(), m.seq_alloc()
        m.seq_forward(s1, ids[:11], 0, want_logits=False); m.seq_forward(s2, ids[:20], 0, want_logits=False)
        lg, _ = m.step_batch_decode([s1, s2], [7, 8])                                   # batched path
        o2.forward(ids[:11], 0); assert rel(lg[0, 0], o2.forward([7], 11)) < 1e-4
        o2.forward(ids[:20], 0); assert rel(lg[1, 0], o2.forward([8], 20)) < 1e-4
    finally:
        m.close()


@pytest.mark.gpu
def test_hip_qwen38_27b_geometry_against_the_hf_golden():
    """The HIP path against HF Qwen3_5ForCausalLM (f32) at the REAL Qwen3.8-27B layer geometry -- H 5120, 24 q / 4 kv heads x 256
    (n_rep 6), 16 key / 48 value GDN heads, I 17408; 4 layers, 4096-entry vocabulary -- on the committed fixture
    tests/golden/qwen3_5_qwen3.8-27b-geom4.npz (make_golden_qwen3_5.py), in the benchmarked KV mode (f16 pages, the default):
    prompt logits through the MFMA prefill, one decode step, and the greedy continuation.  Bar: north_star's 1e-3."""
    from crane_amd.backend import GenerationConfig, Model
    g = np.load(os.path.join(GOLD, "qwen3_5_qwen3.8-27b-geom4.npz"))
    cfg = dict(configs.get_config("qwen3.8-27b"), num_hidden_layers=4, vocab_size=4096, max_position_embeddings=4096)
    m = Model.synthetic(cfg, seed=int(g["seed"][0]), max_seq_len=256, max_seqs=2)
    try:
        ids = g["prompt"].tolist()
        a = m.forward_step(ids, 0)[0, 0]
        assert rel(a, g["prefill_logits"]) < 1e-3, rel(a, g["prefill_logits"])
        b = m.forward_step(g["decode_token"].tolist(), len(ids))[0, 0]
        assert rel(b, g["decode_logits"]) < 1e-3, rel(b, g["decode_logits"])
        # token-serial prompt (every step a decode step: the GEMV / gdn_decode / split-KV kernels) on the same fixture
        m.debug_set("no_prefill", 1)
        try:
            m.clear_kv_cache()
            c = m.forward_step(ids, 0)[0, 0]
        finally:
            m.debug_set("no_prefill", 0)
        assert rel(c, g["prefill_logits"]) < 1e-3, rel(c, g["prefill_logits"])
        want = g["greedy_tokens"].tolist()
        assert m.generate(ids, GenerationConfig.greedy(len(want) - len(ids))) == want
    finally:
        m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("nseq", [9, 17, 64, 96])
def test_batched_decode_at_hidden_5120(nseq):
    """K = H = 5120 > 4096: the SiLU*mul and arg-max projections have no matrix-core GEMV form (gemvm_ok), so groups of more than
    8 sequences reach the VALU batched GEMV, which keeps 8 rows in LDS -- it must run in passes of 8 (9 sequences: gate||up
    and lm_head; 17 / 64: lm_head behind the GEMM projections; 96: the 128-row LDS-DMA GEMM tiles and the lm_head GEMM).  Every row of the batched step against the same sequence
    stepped alone (a fork taken before the step), Qwen3.8-27B layer geometry."""
    from crane_amd.backend import Model
    cfg = dict(configs.get_config("qwen3.8-27b"), num_hidden_layers=4, vocab_size=4096, max_position_embeddings=4096)
    V = cfg["vocab_size"]
    m = Model.synthetic(cfg, seed=0, max_seq_len=128, max_seqs=2 * nseq + 2, kv_dtype="f32")
    try:
        seqs, twins, lens = [], [], []
        for i in range(nseq):
            n = 2 + (5 * i) % 23
            s_ = m.seq_alloc()
            m.seq_forward(s_, [(13 * i + 7 * k + 3) % V for k in range(n)], 0, want_logits=False)
            seqs.append(s_); twins.append(m.seq_fork(s_)); lens.append(n)
        toks = [(5 + 3 * i) % V for i in range(nseq)]
        lg, greedy = m.step_batch_decode(seqs, toks)
        for i in range(nseq):
            ref, gr = m.seq_forward(twins[i], [toks[i]], lens[i])
            assert rel(lg[i, 0], ref.reshape(-1)) < 1e-4, (i, rel(lg[i, 0], ref.reshape(-1)))
            assert int(greedy[i]) == int(lg[i, 0].argmax())
    finally:
        m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kv", ["f16", "bf16", "int8"])
def test_hybrid_decode_attention_on_the_mfma_kernel(kv):
    """head_dim 256 gated attention of the hybrid family through the matrix-core flash-decode kernel (single register
    set; the sigmoid gate stays in the combine kernel), forced from the first token and with the 32 -> 64 split switch."""
    from crane_amd import configs, synth
    from crane_amd.backend import Model
    from oracle import qwen3_5_oracle as O5
    cfg = configs.get_config("tiny-qwen3.5")
    w = synth.synth_weights_f32(cfg, seed=0)
    o = O5.Qwen35Oracle(O5.Qwen35Config.from_json(cfg), w, kv_dtype=kv)
    m = Model.synthetic(cfg, seed=0, max_seq_len=512, max_seqs=3, kv_dtype=kv)
    tol = 2e-3 if kv == "int8" else 5e-4
    try:
        m.debug_set("attn_mfma_min", 1)
        m.debug_set("attn_mfma_wide_min", 150)
        ids = configs.synthetic_prompt(140, cfg["vocab_size"])
        ref = o.forward(ids, 0)
        assert np.abs(m.forward_step(ids, 0).reshape(-1) - ref).max() / np.abs(ref).max() < tol
        tok = int(ref.argmax())
        for step in range(16):
            ref = o.forward([tok], 140 + step)
            got = m.forward_step([tok], 140 + step).reshape(-1)
            assert np.abs(got - ref).max() / np.abs(ref).max() < tol, step
            tok = int(ref.argmax())
    finally:
        m.close()


@pytest.mark.gpu
def test_gated_norm_in_the_out_proj_prologue_equals_the_in_step_norm():
    """Single-sequence decode: the Gated-Delta-Net step ends at its raw y and the out_proj GEMV applies the gated RMSNorm of each
    128-wide value head while it stages x (PRO_GDNNORM, the default) -- against the step that normalises itself through the
    last-arriving workgroup (cm_debug_set("gdn_defer_norm", 0)): same logits up to the association of the 128-term sum of squares,
    same greedy tokens; and against the oracle."""
    from crane_amd import configs, synth
    from crane_amd.backend import GenerationConfig, Model
    from oracle import qwen3_5_oracle as O5
    cfg = configs.get_config("tiny-qwen3.5")
    o = O5.Qwen35Oracle(O5.Qwen35Config.from_json(cfg), synth.synth_weights_f32(cfg, seed=0))
    m = Model.synthetic(cfg, seed=0, max_seq_len=256, max_seqs=2, kv_dtype="f32")
    try:
        ids = configs.synthetic_prompt(19, cfg["vocab_size"])
        outs = []
        for defer in (1, 0):
            m.debug_set("gdn_defer_norm", defer)
            m.clear_kv_cache()
            m.forward_step(ids, 0)
            lg = [m.forward_step([7 + i], 19 + i)[0, 0].copy() for i in range(4)]
            m.clear_kv_cache()
            toks = m.generate(ids, GenerationConfig.greedy(12))
            outs.append((lg, toks))
        o.forward(ids, 0)
        for i in range(4):
            ref = o.forward([7 + i], 19 + i)
            assert rel(outs[0][0][i], ref) < 1e-4, (i, rel(outs[0][0][i], ref))
            assert rel(outs[0][0][i], outs[1][0][i]) < 1e-5, (i, rel(outs[0][0][i], outs[1][0][i]))
        assert outs[0][1] == outs[1][1]
    finally:
        m.debug_set("gdn_defer_norm", 1)
        m.close()


@pytest.mark.gpu
def test_hybrid_per_layer_chain_equals_the_launch_path():
    """Qwen3.8-27B layer geometry (hidden 5120, intermediate 17 408, 48 value heads / 24 gated q heads of 256): decode on the
    persistent per-layer chain (out_proj / o_proj -> gate||up -> down_proj -> the next layer's in_proj / QKV in ONE launch, 1024-element
    dependency chunks, 17 chunks for down_proj; the Gated-Delta-Net step and the attention stay launches) against the
    per-projection launch path: logits to summation order, greedy tokens equal -- with the graph-replay loop as the first use of
    the handle."""
    from crane_amd import configs
    from crane_amd.backend import GenerationConfig, Model
    cfg = dict(configs.get_config("qwen3.8-27b"), num_hidden_layers=8, vocab_size=4096, max_position_embeddings=4096)
    outs = []
    for engine in (1, -1):
        m = Model.synthetic(cfg, seed=0, max_seq_len=2048, max_seqs=2, engine=engine)
        try:
            assert m.engine_active() == (1 if engine == 1 else 0)
            m.debug_fill_kv(900, seed=2)
            toks, _ = m.bench_decode(3, 10)
            assert m.engine_active() == (1 if engine == 1 else 0)
            m.clear_kv_cache()
            ids = configs.synthetic_prompt(23, cfg["vocab_size"])
            m.forward_step(ids, 0)
            lg = [m.forward_step([11 + i], 23 + i)[0, 0].copy() for i in range(3)]
            m.clear_kv_cache()
            gen = m.generate(ids, GenerationConfig.greedy(10))
            outs.append(([int(t) for t in toks], lg, gen))
        finally:
            m.close()
    assert outs[0][0] == outs[1][0] and outs[0][2] == outs[1][2]
    for a, b in zip(outs[0][1], outs[1][1]):
        assert rel(a, b) < 1e-4, rel(a, b)
