"""GPU parity: HIP path (through the C ABI) vs the CPU oracle, Qwen3 dense.

Tolerances (BASELINE.json north_star): logits within 1e-3 relative
(max|d| / max|ref|) of the CPU f32 forward on identical bf16-rounded weights;
greedy token ids bit-exact.  See REL_* below for what is asserted in each KV mode.
"""
import numpy as np
import pytest

from crane_amd import configs, synth
from crane_amd.backend import GenerationConfig, ListStreamer, Model
from oracle.qwen3_oracle import Qwen3Config, Qwen3Oracle

pytestmark = pytest.mark.gpu

REL_SAME = 5e-4     # vs the oracle with the same KV rounding: two summation orders put a few K/V elements on opposite
                    # sides of a rounding tie (bf16 pages: 2^-9 on those elements, measured <= 2.6e-4 on logits; f16
                    # pages: 2^-12).  The f32-KV tests below have no such flips and hold 1e-4.
# vs the PURE-f32 CPU forward (the reference's CPU path; BASELINE.json north_star: 1e-3 relative):
#   f16 pages (the default and the benchmarked mode) are held to the north-star bar itself;
#   bf16 pages (opt-in: the model dtype of the reference's GPU path) carry one bf16 epsilon -- a 1-token context returns
#   bf16(v) exactly, 2^-9 on every element -- and are NOT claimed to meet it.
REL_F32 = {"f16": 1e-3, "bf16": 4e-3}


def rel(a, ref):
    return float(np.abs(a - ref).max() / np.abs(ref).max())


@pytest.fixture(scope="module", params=[("tiny-qwen3", "f16"), ("tiny-qwen3-untied", "f16"), ("tiny-qwen3", "bf16")],
                ids=lambda p: f"{p[0]}-{p[1]}")
def pair(request):
    name, kv = request.param
    cfg = configs.get_config(name)
    w = synth.synth_weights_f32(cfg, seed=0)
    m = Model.synthetic(cfg, seed=0, max_seq_len=512, max_seqs=4, kv_dtype=kv)
    m.kv = kv                                   # the oracle of a test rounds K/V at the same point
    yield cfg, w, m
    m.close()


def test_decode_logits_match_oracle(pair):
    cfg, w, m = pair
    c = Qwen3Config.from_json(cfg)
    o_same = Qwen3Oracle(c, w, kv_dtype=m.kv)
    o_f32 = Qwen3Oracle(c, w)
    ids = configs.synthetic_prompt(24, cfg["vocab_size"])
    m.clear_kv_cache()
    for pos, t in enumerate(ids):
        got = m.forward_step([t], pos)
        assert got.shape == (1, 1, cfg["vocab_size"])
        a = o_same.forward([t], pos)
        b = o_f32.forward([t], pos)
        assert rel(got[0, 0], a) < REL_SAME, (pos, rel(got[0, 0], a))
        assert rel(got[0, 0], b) < REL_F32[m.kv], (pos, rel(got[0, 0], b))
        assert int(got[0, 0].argmax()) == int(b.argmax())


def test_f32_kv_meets_north_star_bar():
    """kv_dtype=f32: logits within 1e-3 relative of the pure-f32 CPU forward (measured ~1e-6)."""
    cfg = configs.get_config("tiny-qwen3-untied")
    w = synth.synth_weights_f32(cfg, seed=0)
    o = Qwen3Oracle(Qwen3Config.from_json(cfg), w)
    m = Model.synthetic(cfg, seed=0, max_seq_len=256, max_seqs=2, kv_dtype="f32")
    try:
        ids = configs.synthetic_prompt(20, cfg["vocab_size"])
        worst = 0.0
        for pos, t in enumerate(ids):
            worst = max(worst, rel(m.forward_step([t], pos)[0, 0], o.forward([t], pos)))
        assert worst < 1e-4, worst
    finally:
        m.close()


def test_prompt_forward_and_greedy(pair):
    cfg, w, m = pair
    o = Qwen3Oracle(Qwen3Config.from_json(cfg), w, kv_dtype=m.kv)
    ids = configs.synthetic_prompt(33, cfg["vocab_size"])
    m.clear_kv_cache()
    got = m.forward_step(ids, 0)[0, 0]
    ref = o.forward(ids, 0)
    assert rel(got, ref) < REL_SAME
    m.clear_kv_cache()
    assert m.forward_step_greedy(ids, 0) == int(ref.argmax())


def test_generate_tokens_bit_exact(pair):
    cfg, w, m = pair
    o = Qwen3Oracle(Qwen3Config.from_json(cfg), w)          # pure f32 reference forward
    ids = configs.synthetic_prompt(9, cfg["vocab_size"])
    ref, logits = o.generate(ids, 16, return_logits=True)
    # margins must dwarf the bf16-KV noise for the equality to be meaningful
    margins = [np.sort(l)[-1] - np.sort(l)[-2] for l in logits]
    assert min(margins) > 1e-3 * max(np.abs(l).max() for l in logits)
    st = ListStreamer()
    got = m.generate(ids, GenerationConfig.greedy(16), st)
    assert got == ref
    assert st.tokens == ref[len(ids):] and st.finalized
    got8 = m.generate(ids, GenerationConfig.greedy(16), sync_every=8)
    assert got8 == ref


def test_generate_eos_and_repeat_penalty(pair):
    cfg, w, m = pair
    o = Qwen3Oracle(Qwen3Config.from_json(cfg), w)
    ids = configs.synthetic_prompt(5, cfg["vocab_size"])
    ref = o.generate(ids, 12)
    eos = ref[len(ids) + 3]
    stop_at = ref.index(eos, len(ids)) + 1
    got = m.generate(ids, GenerationConfig.greedy(12, eos_token_id=eos), sync_every=4)
    assert got == ref[:stop_at]                             # EOS token is pushed, then stop (model.rs:318-327)
    refp = o.generate(ids, 10, repetition_penalty=1.3, repeat_last_n=4)
    gc = GenerationConfig.greedy(10)
    gc.repetition_penalty, gc.repeat_last_n = 1.3, 4
    assert m.generate(ids, gc) == refp


def test_from_pretrained_equals_synthetic(pair, tmp_path):
    cfg, w, m = pair
    d = synth.write_model_dir(str(tmp_path / "ckpt"), cfg, seed=0, shards=2)
    m2 = Model.from_pretrained(d, max_seq_len=256, max_seqs=2, kv_dtype=m.kv)
    try:
        ids = configs.synthetic_prompt(7, cfg["vocab_size"])
        m.clear_kv_cache()
        a = m.forward_step(ids, 0)
        b = m2.forward_step(ids, 0)
        assert np.array_equal(a, b)                          # same kernels, same bits
        assert m2.num_layers() == cfg["num_hidden_layers"]
    finally:
        m2.close()


def test_sequences_fork_truncate(pair):
    cfg, w, m = pair
    o = Qwen3Oracle(Qwen3Config.from_json(cfg), w, kv_dtype=m.kv)
    V = cfg["vocab_size"]
    a = configs.synthetic_prompt(70, V)                      # crosses a 64-token page
    s1 = m.seq_alloc()
    la, _ = m.seq_forward(s1, a, 0)
    s2 = m.seq_fork(s1)
    assert m.seq_len(s2) == 70
    l1, _ = m.seq_forward(s1, [11], 70)
    l2, _ = m.seq_forward(s2, [23], 70)                      # diverge after the fork
    o.clear_kv_cache(); o.forward(a, 0); r1 = o.forward([11], 70)
    o.clear_kv_cache(); o.forward(a, 0); r2 = o.forward([23], 70)
    assert rel(l1, r1) < REL_SAME and rel(l2, r2) < REL_SAME
    m.seq_truncate(s1, 40)                                   # preemption: drop suffix, re-decode
    l3, _ = m.seq_forward(s1, a[40:50], 40)
    o.clear_kv_cache(); o.forward(a[:40], 0); r3 = o.forward(a[40:50], 40)
    assert rel(l3, r3) < REL_SAME
    lg, g = m.step_batch_decode([s1, s2], [5, 6])
    assert lg.shape == (2, 1, V) and g.shape == (2,)
    m.seq_free(s1); m.seq_free(s2)
    assert m.active_kv_cache_bytes() == 0 or m.seq_len(0) > 0


def test_fork_truncate_append_does_not_touch_the_sibling(pair):
    """A fork cut back INTO a page it still shares with its parent (truncate, or forward with start_pos < len) must get a
    private copy of that page before it appends: fork at 160 tokens (2.5 pages), truncate the child to 100 (inside the
    fully shared second page), re-feed other tokens -- the parent's positions 100..127 must be unchanged."""
    cfg, w, m = pair
    o = Qwen3Oracle(Qwen3Config.from_json(cfg), w, kv_dtype=m.kv)
    V = cfg["vocab_size"]
    a = configs.synthetic_prompt(160, V)
    parent = m.seq_alloc()
    m.seq_forward(parent, a, 0)
    child = m.seq_fork(parent)
    other = [(5 * i + 1) % V for i in range(40)]
    lc, _ = m.seq_forward(child, other, 100)                 # start_pos < len: truncates to 100, then appends 40 tokens
    lp, _ = m.seq_forward(parent, [9], 160)                  # the parent still attends to ITS tokens 100..159
    o.clear_kv_cache(); o.forward(a, 0); rp = o.forward([9], 160)
    o.clear_kv_cache(); o.forward(a[:100], 0); rc = o.forward(other, 100)
    assert rel(lp, rp) < REL_SAME, rel(lp, rp)
    assert rel(lc, rc) < REL_SAME, rel(lc, rc)
    m.seq_truncate(parent, 70)                               # and the other way round: the parent is cut into a shared page
    lp2, _ = m.seq_forward(parent, [3, 4, 5], 70)
    lc2, _ = m.seq_forward(child, [8], 140)
    o.clear_kv_cache(); o.forward(a[:70], 0); rp2 = o.forward([3, 4, 5], 70)
    o.clear_kv_cache(); o.forward(a[:100], 0); o.forward(other, 100); rc2 = o.forward([8], 140)
    assert rel(lp2, rp2) < REL_SAME and rel(lc2, rc2) < REL_SAME
    m.seq_free(parent); m.seq_free(child)


def test_error_behaviour(pair):
    cfg, w, m = pair
    from crane_amd._lib import CraneError
    m.clear_kv_cache()
    with pytest.raises(CraneError):
        m.forward_step([cfg["vocab_size"]], 0)               # token id out of range
    with pytest.raises(CraneError):
        m.forward_step([1], 5)                               # start_pos beyond cache
    with pytest.raises(CraneError):
        m.forward_step([], 0)
    with pytest.raises(CraneError):
        Model.synthetic(dict(cfg, model_type="llama"))
    with pytest.raises(CraneError):
        m.generate([1, 2], GenerationConfig.with_max_tokens(10 ** 6))   # prompt + max_new_tokens > max_seq_len
    out = m.generate([1, 2], GenerationConfig.with_max_tokens(4))       # default config samples (temperature 0.67)
    assert len(out) == 6 and out[:2] == [1, 2]


# ---------------------------------------------------------------------------------------------
# prefill (MFMA GEMM + flash attention) path
# ---------------------------------------------------------------------------------------------
def _serial(m, ids, start):
    m.debug_set("no_prefill", 1)
    try:
        return m.forward_step(ids, start)[0, 0]
    finally:
        m.debug_set("no_prefill", 0)


@pytest.mark.parametrize("n", [2, 17, 64, 65, 200])
def test_prefill_matches_token_serial_and_oracle(pair, n):
    """MFMA prefill == token-by-token GEMV path == oracle (ragged / page-crossing lengths)."""
    cfg, w, m = pair
    o = Qwen3Oracle(Qwen3Config.from_json(cfg), w, kv_dtype=m.kv)
    ids = configs.synthetic_prompt(n, cfg["vocab_size"])
    m.clear_kv_cache()
    a = m.forward_step(ids, 0)[0, 0]
    m.clear_kv_cache()
    b = _serial(m, ids, 0)
    ref = o.forward(ids, 0)
    assert rel(a, b) < REL_SAME, rel(a, b)
    assert rel(a, ref) < REL_SAME, rel(a, ref)
    # decode continues correctly from a prefilled cache
    nxt = m.forward_step([5], n)[0, 0]
    assert rel(nxt, o.forward([5], n)) < REL_SAME


def test_chunked_prefill_matches_single(pair):
    """reference KAT qwen3/modeling.rs:1763-1801 (chunked vs single prefill, tol 1e-4), at page-crossing sizes."""
    cfg, w, m = pair
    ids = configs.synthetic_prompt(150, cfg["vocab_size"])
    m.clear_kv_cache()
    m.forward_step(ids, 0)
    single = m.forward_step([6], 150)[0, 0]
    m.clear_kv_cache()
    m.forward_step(ids[:70], 0)
    m.forward_step(ids[70:], 70)            # second chunk attends to the cached prefix (kv_offset > 0)
    chunked = m.forward_step([6], 150)[0, 0]
    assert rel(single, chunked) < 1e-4
    m2 = Model.synthetic(cfg, seed=0, max_seq_len=512, max_seqs=2, prefill_chunk=48, kv_dtype=m.kv)   # internal chunk loop
    try:
        m2.forward_step(ids, 0)
        assert rel(m2.forward_step([6], 150)[0, 0], single) < 1e-4
    finally:
        m2.close()


def test_prefill_f32_kv_and_plain_bf16_modes():
    cfg = configs.get_config("tiny-qwen3-untied")
    w = synth.synth_weights_f32(cfg, seed=0)
    o = Qwen3Oracle(Qwen3Config.from_json(cfg), w)                 # pure f32 CPU forward
    ids = configs.synthetic_prompt(96, cfg["vocab_size"])
    ref = o.forward(ids, 0)
    m = Model.synthetic(cfg, seed=0, max_seq_len=256, max_seqs=2, kv_dtype="f32")
    try:
        assert rel(m.forward_step(ids, 0)[0, 0], ref) < 1e-4        # north-star bar is 1e-3
    finally:
        m.close()
    m = Model.synthetic(cfg, seed=0, max_seq_len=256, max_seqs=2, prefill_split=1)   # plain bf16 activations
    try:
        got = m.forward_step(ids, 0)[0, 0]
        assert rel(got, ref) < 2e-2 and int(got.argmax()) == int(ref.argmax())
    finally:
        m.close()


@pytest.mark.parametrize("tp_graph", ["1", "0"])
def test_rccl_code_path_single_rank(tp_graph):
    """cm_opts.debug_flags = CM_DEBUG_FORCE_RCCL: the tp=1 reductions go through a 1-rank RCCL communicator (dlopen, unique-id ABI,
    all-reduce + all-gather on the model's stream), captured into the decode hipGraph (CM_TP_GRAPH=1, the default) or
    launched eagerly (0).  Results must equal the collective-free path."""
    cfg = configs.get_config("tiny-qwen3-untied")
    ids = configs.synthetic_prompt(40, cfg["vocab_size"])
    def batched(m):        # cm_decode_batch: 3 sequences (matrix-core GEMVs), all-reduce of [3, H] + in-place gathers
        seqs = [0, m.seq_alloc(), m.seq_alloc()]
        for i, s in enumerate(seqs):
            m.seq_forward(s, ids[: 9 + 4 * i], 0, want_logits=False)
        return m.step_batch_decode(seqs, [7, 8, 9])
    m = Model.synthetic(cfg, seed=0, max_seq_len=256, max_seqs=4)
    try:
        a = m.forward_step(ids, 0)[0, 0]
        a2 = m.forward_step([7], 40)[0, 0]
        a3, g3 = batched(m)
    finally:
        m.close()
    m = Model.synthetic(cfg, seed=0, max_seq_len=256, max_seqs=4, debug_force_rccl=True)
    try:
        m.debug_set("tp_graph", int(tp_graph))
        b = m.forward_step(ids, 0)[0, 0]
        b2 = m.forward_step([7], 40)[0, 0]
        b3, h3 = batched(m)
        toks = m.generate(ids[:5], GenerationConfig.greedy(6), sync_every=3)
    finally:
        m.close()
    assert rel(b, a) < 1e-6 and rel(b2, a2) < 1e-6 and len(toks) == 11
    assert rel(b3, a3) < 1e-6 and list(h3) == list(g3)


@pytest.mark.parametrize("nseq", [2, 3, 8, 11, 16, 20, 32, 41, 64])
def test_batched_decode_matches_per_sequence_oracle(pair, nseq):
    """step_batch_decode (backend.rs:107-121): N sequences of different lengths, one token each, ONE pass over the
    weights (gemvb kernels) -- no padding / masks; every row must equal that sequence's own single-step result."""
    cfg, w, m = pair
    if nseq > 3:
        m = Model.synthetic(cfg, seed=0, max_seq_len=512, max_seqs=66, kv_dtype="f32")
    V = cfg["vocab_size"]
    try:
        oracles, seqs, lens = [], [], []
        for i in range(nseq):
            n = 3 + 5 * i + (60 if i == 1 else 0)              # ragged lengths, one crossing a page
            ids = [(11 * i + 7 * k + 3) % V for k in range(n)]
            o = Qwen3Oracle(Qwen3Config.from_json(cfg), w, kv_dtype=m.kv if nseq <= 3 else "f32")
            o.forward(ids, 0)
            s = m.seq_alloc()
            m.seq_forward(s, ids, 0, want_logits=False)
            oracles.append(o); seqs.append(s); lens.append(n)
        toks = [(5 + i) % V for i in range(nseq)]
        for step in range(2):
            lg, greedy = m.step_batch_decode(seqs, toks)
            assert lg.shape == (nseq, 1, V)
            nxt = []
            for i in range(nseq):
                ref = oracles[i].forward([toks[i]], lens[i] + step)
                assert rel(lg[i, 0], ref) < (REL_SAME if nseq <= 3 else 1e-4), (i, step)
                assert int(greedy[i]) == int(lg[i, 0].argmax())
                nxt.append(int(ref.argmax()))
            toks = nxt
        for s in seqs:
            assert m.seq_len(s) == lens[seqs.index(s)] + 2
            m.seq_free(s)
    finally:
        if nseq > 3:
            m.close()


@pytest.mark.parametrize("nseq", [9, 12, 17, 40, 64, 65, 100, 128])
def test_batched_decode_gemm_path_against_gemv_path(nseq):
    """From 9 sequences on (batch_gemm_min) cm_decode_batch runs the projections as MFMA GEMMs over the rows of the batch (M = nseq,
    split-K) instead of the batched matrix-core GEMVs.  The same step through both paths (the step is taken back with
    cm_seq_truncate in between): logits agree to the summation order of two bf16 hi + lo products, tokens are equal."""
    cfg = configs.get_config("tiny-qwen3-untied")
    V = cfg["vocab_size"]
    m = Model.synthetic(cfg, seed=0, max_seq_len=256, max_seqs=nseq + 2, kv_dtype="f32")
    try:
        seqs, lens = [], []
        for i in range(nseq):
            n = 2 + (7 * i) % 45
            s = m.seq_alloc()
            m.seq_forward(s, [(13 * i + 5 * k + 1) % V for k in range(n)], 0, want_logits=False)
            seqs.append(s); lens.append(n)
        toks = [(3 + 2 * i) % V for i in range(nseq)]
        for step in range(2):
            m.debug_set("batch_gemm_min", 0)
            lg_v, gr_v = m.step_batch_decode(seqs, toks)
            for s_, n_ in zip(seqs, lens):
                m.seq_truncate(s_, n_ + step)              # take the step back: the GEMM path appends the same K / V rows
            m.debug_set("batch_gemm_min", 9)
            lg_m, gr_m = m.step_batch_decode(seqs, toks)
            assert not np.array_equal(lg_m, lg_v)          # (the other path did run)
            for i in range(nseq):
                assert rel(lg_m[i, 0], lg_v[i, 0]) < 1e-4, (step, i)
                assert m.seq_len(seqs[i]) == lens[i] + step + 1
            assert [int(t) for t in gr_m] == [int(t) for t in gr_v]
            toks = [int(t) for t in gr_m]
    finally:
        m.close()


@pytest.mark.parametrize("name", ["tiny-qwen3-untied", "tiny-qwen3.5"])
@pytest.mark.parametrize("heads_max,ns", [(0, 2), (4096, 1), (4096, 2), (4096, 4), (150, 2)])
def test_decode_attention_variants(name, heads_max, ns):
    """The two decode-attention formulations (split-KV + combine launch; per-head blocks with the merge fused into
    o_proj's prologue) and the switch between them at a context threshold must all reproduce the oracle -- single
    sequence (graph per variant) and batched, over a context that spans several pages."""
    cfg = configs.get_config(name)
    w = synth.synth_weights_f32(cfg, seed=0)
    if name == "tiny-qwen3.5":
        from oracle import qwen3_5_oracle as O5
        o = O5.Qwen35Oracle(O5.Qwen35Config.from_json(cfg), w)
        o2 = O5.Qwen35Oracle(O5.Qwen35Config.from_json(cfg), w)
    else:
        o = Qwen3Oracle(Qwen3Config.from_json(cfg), w)
        o2 = Qwen3Oracle(Qwen3Config.from_json(cfg), w)
    m = Model.synthetic(cfg, seed=0, max_seq_len=512, max_seqs=3, kv_dtype="f32")
    try:
        m.debug_set("attn_heads_max", heads_max)
        m.debug_set("attn_ns", ns)
        V = cfg["vocab_size"]
        ids = configs.synthetic_prompt(140, V)
        ref = o.forward(ids, 0)
        assert rel(m.forward_step(ids, 0).reshape(-1), ref) < 1e-4
        tok = int(ref.argmax())
        for step in range(14):                                   # crosses 150 tokens of context
            ref = o.forward([tok], 140 + step)
            got = m.forward_step([tok], 140 + step).reshape(-1)
            assert rel(got, ref) < 1e-4, step
            tok = int(ref.argmax())
        # batched: the same context in two extra sequences, one of them shorter
        s1, s2 = m.seq_alloc(), m.seq_alloc()
        m.seq_forward(s1, ids, 0, want_logits=False)
        m.seq_forward(s2, ids[:33], 0, want_logits=False)
        o.forward(ids, 0); o2.forward(ids[:33], 0)
        t1, t2 = 5, 9
        for step in range(12):
            lg, _ = m.step_batch_decode([s1, s2], [t1, t2])
            r1, r2 = o.forward([t1], 140 + step), o2.forward([t2], 33 + step)
            assert rel(lg[0].reshape(-1), r1) < 1e-4 and rel(lg[1].reshape(-1), r2) < 1e-4, step
            t1, t2 = int(r1.argmax()), int(r2.argmax())
    finally:
        m.close()


@pytest.mark.parametrize("kv", ["f16", "bf16"])
@pytest.mark.parametrize("mfma_min,wide_min", [(1, 8192), (200, 206), (1, 1)])
def test_decode_attention_mfma_variant(mfma_min, wide_min, kv):
    """Long-context decode attention on the matrix cores (f16 / bf16 KV, head_dim 128): S^T = K.Q^T over the GQA group and
    O^T += V^T.P^T through ds_read_tr, same partial format / combine kernel.  Forced on from the first token (1) and
    switched on mid-generation (200): must match the oracle with the same KV rounding, single sequence and batched."""
    for name in ("tiny-qwen3-untied", "tiny-qwen3"):
        cfg = configs.get_config(name)
        w = synth.synth_weights_f32(cfg, seed=0)
        o = Qwen3Oracle(Qwen3Config.from_json(cfg), w, kv_dtype=kv)
        o2 = Qwen3Oracle(Qwen3Config.from_json(cfg), w, kv_dtype=kv)
        m = Model.synthetic(cfg, seed=0, max_seq_len=512, max_seqs=3, kv_dtype=kv)
        try:
            m.debug_set("attn_mfma_min", mfma_min)
            m.debug_set("attn_mfma_wide_min", wide_min)          # 32 -> 64 token splits (a third captured graph)
            V = cfg["vocab_size"]
            ids = configs.synthetic_prompt(190, V)
            ref = o.forward(ids, 0)
            assert rel(m.forward_step(ids, 0).reshape(-1), ref) < REL_SAME
            tok = int(ref.argmax())
            for step in range(24):                                   # crosses 200 tokens and several 16/64-token tiles
                ref = o.forward([tok], 190 + step)
                got = m.forward_step([tok], 190 + step).reshape(-1)
                assert rel(got, ref) < REL_SAME, (name, step, rel(got, ref))
                assert int(got.argmax()) == int(ref.argmax())
                tok = int(ref.argmax())
            s1, s2 = m.seq_alloc(), m.seq_alloc()
            m.seq_forward(s1, ids, 0, want_logits=False)
            m.seq_forward(s2, ids[:77], 0, want_logits=False)
            o.forward(ids, 0); o2.forward(ids[:77], 0)
            t1, t2 = 5, 9
            for step in range(12):
                lg, _ = m.step_batch_decode([s1, s2], [t1, t2])
                r1, r2 = o.forward([t1], 190 + step), o2.forward([t2], 77 + step)
                assert rel(lg[0].reshape(-1), r1) < REL_SAME and rel(lg[1].reshape(-1), r2) < REL_SAME, (name, step)
                t1, t2 = int(r1.argmax()), int(r2.argmax())
        finally:
            m.close()


@pytest.mark.parametrize("name", ["tiny-qwen3-untied", "eng-qwen3"])
def test_f16_pages_saturate_like_the_oracle_when_a_value_row_overflows(name, tmp_path):
    """CM_KV_F16 pages hold |x| <= 65504.  A checkpoint whose layer-0 v_proj is scaled by 2^16 (exact in bf16) drives V rows far
    past that: the append must CLAMP (v_cvt_f16_f32 after a clamp, dev_common.h) -- never write inf, which would turn the softmax
    average into NaN -- exactly like the oracle's f16 rounding point (oracle/qwen3_oracle.py f16_round, qc_common.h f16_round).
    Covered appends: the prompt pass (qknorm_rope_kv_kernel), the split-KV decode step, and -- eng-qwen3 -- the persistent decode
    kernel's in-kernel append."""
    import json, os
    cfg = configs.get_config(name)
    d = str(tmp_path / "ckpt")
    os.makedirs(d)
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(cfg, f)
    w, tensors = {}, []
    for tname, shape, std, off in synth.specs_for(cfg):
        bits = synth.synth_bf16_bits(tname, int(np.prod(shape)), 0, std, off)
        if tname == "model.layers.0.self_attn.v_proj.weight":
            bits = synth.f32_to_bf16_bits(synth.bf16_bits_to_f32(bits) * np.float32(2.0 ** 16))
        w[tname] = synth.bf16_bits_to_f32(bits).reshape(shape)
        tensors.append((tname, shape, bits))
    synth.write_safetensors_bf16(os.path.join(d, "model.safetensors"), tensors)
    c = Qwen3Config.from_json(cfg)
    o = Qwen3Oracle(c, w, kv_dtype="f16")
    V = cfg["vocab_size"]
    ids = configs.synthetic_prompt(40, V)
    m = Model.from_pretrained(d, max_seq_len=256, max_seqs=2)            # default options: f16 pages
    try:
        ref = o.forward(ids, 0)
        assert np.abs(o.v_cache[0]).max() == 65504.0                      # the oracle's cache really saturated
        got = m.forward_step(ids, 0)[0, 0]
        assert np.isfinite(got).all()
        assert rel(got, ref) < REL_SAME, rel(got, ref)
        tok = int(ref.argmax())
        for step in range(3):                                             # decode steps append (and clamp) their own rows
            ref = o.forward([tok], 40 + step)
            got = m.forward_step([tok], 40 + step)[0, 0]
            assert np.isfinite(got).all()
            assert rel(got, ref) < REL_SAME, (step, rel(got, ref))
            tok = int(ref.argmax())
        m.debug_set("no_prefill", 1)                                      # the prompt token by token: every append is a decode step
        m.clear_kv_cache()
        o2 = Qwen3Oracle(c, w, kv_dtype="f16")
        got = m.forward_step(ids[:12], 0)[0, 0]
        assert rel(got, o2.forward(ids[:12], 0)) < REL_SAME
    finally:
        m.close()
