"""GPU: int8 / int4 per-token quantised KV cache (KvCache::Quant, qwen3_5/kv_cache.rs:209-342) with the dequantisation
fused into the decode-attention kernel, against the oracle that quantises/dequantises at the same points."""
import numpy as np
import pytest

from crane_amd import configs, synth

pytestmark = pytest.mark.gpu


def rel(a, ref):
    return float(np.abs(a - ref).max() / np.abs(ref).max())


def _oracle(name, cfg, w, kv):
    if name == "tiny-qwen3.5":
        from oracle import qwen3_5_oracle as O5
        return O5.Qwen35Oracle(O5.Qwen35Config.from_json(cfg), w, kv_dtype=kv)
    from oracle.qwen3_oracle import Qwen3Config, Qwen3Oracle
    return Qwen3Oracle(Qwen3Config.from_json(cfg), w, kv_dtype=kv)


@pytest.mark.parametrize("name", ["tiny-qwen3-untied", "tiny-qwen3.5"])
@pytest.mark.parametrize("kv", ["int8", "int4"])
def test_quantised_kv_matches_oracle(name, kv):
    from crane_amd.backend import Model
    cfg = configs.get_config(name)
    w = synth.synth_weights_f32(cfg, seed=0)
    o = _oracle(name, cfg, w, kv)
    m = Model.synthetic(cfg, seed=0, max_seq_len=256, max_seqs=3, kv_dtype=kv)
    try:
        V = cfg["vocab_size"]
        ids = configs.synthetic_prompt(70, V)                       # crosses a 64-token page
        # a code that lands on a rounding tie can flip by one step: compare loosely on logits, exactly on most tokens
        tol = 2e-3 if kv == "int8" else 1e-3
        ref = o.forward(ids, 0)
        got = m.forward_step(ids, 0).reshape(-1)
        assert rel(got, ref) < tol, rel(got, ref)
        tok = int(ref.argmax())
        worst = 0.0
        for step in range(10):
            ref = o.forward([tok], 70 + step)
            got = m.forward_step([tok], 70 + step).reshape(-1)
            worst = max(worst, rel(got, ref))
            tok = int(ref.argmax())
        assert worst < tol, worst
        # KV bytes: codes + one f32 scale per (token, kv head) -- smaller than bf16 by ~2x / ~4x
        mb = Model.synthetic(cfg, seed=0, max_seq_len=256, max_seqs=3)
        try:
            mb.forward_step(ids, 0)
            ratio = mb.active_kv_cache_bytes() / m.active_kv_cache_bytes()
            assert (1.8 < ratio < 2.0) if kv == "int8" else (3.2 < ratio < 4.0), ratio
        finally:
            mb.close()
        # fork shares pages (codes + scales travel together); batched decode over two sequences
        s1 = m.seq_fork(0)
        lg, _ = m.step_batch_decode([0, s1], [5, 5])
        assert rel(lg[0].reshape(-1), lg[1].reshape(-1)) < 2e-5        # same inputs, different MFMA columns / reduction slots
        ref = o.forward([5], 80)
        assert rel(lg[0].reshape(-1), ref) < tol
    finally:
        m.close()


@pytest.mark.parametrize("name", ["tiny-qwen3-untied", "tiny-qwen3.5"])
@pytest.mark.parametrize("kv", ["int8", "int4"])
def test_quantised_kv_chunked_prefill_and_serial_agree(monkeypatch, name, kv):
    """Prompts over a quantised cache: the append kernel quantises K/V and the chunk's attention reads the dequantised
    f32 shadow (old tokens re-dequantised from the pages); chunked (3 chunks), single-chunk and token-serial prefill must
    agree with each other and with the oracle."""
    from crane_amd.backend import Model
    cfg = configs.get_config(name)
    w = synth.synth_weights_f32(cfg, seed=0)
    o = _oracle(name, cfg, w, kv)
    ids = configs.synthetic_prompt(150, cfg["vocab_size"])
    ref = o.forward(ids, 0)
    tol = 2e-3 if kv == "int8" else 1e-3
    outs = []
    for chunk in (0, 64):
        m = Model.synthetic(cfg, seed=0, max_seq_len=256, kv_dtype=kv, prefill_chunk=chunk)
        try:
            outs.append(m.forward_step(ids, 0).reshape(-1).copy())
            nxt = m.forward_step([int(ref.argmax())], 150).reshape(-1).copy()       # decode on the pages the prefill wrote
            outs.append(nxt)
        finally:
            m.close()
    m = Model.synthetic(cfg, seed=0, max_seq_len=256, kv_dtype=kv)
    try:
        m.debug_set("no_prefill", 1)
        serial = m.forward_step(ids, 0).reshape(-1).copy()
    finally:
        m.close()
    ref2 = o.forward([int(ref.argmax())], 150)
    assert rel(outs[0], ref) < tol and rel(outs[2], ref) < tol and rel(serial, ref) < tol, (rel(outs[0], ref), rel(outs[2], ref), rel(serial, ref))
    assert rel(outs[1], ref2) < tol and rel(outs[3], ref2) < tol


def test_quantised_kv_is_close_to_full_precision():
    """int8 KV must stay within ~1e-2 of the f32-KV logits on the same weights (sanity of the scales)."""
    from crane_amd.backend import Model
    cfg = configs.get_config("tiny-qwen3-untied")
    ids = configs.synthetic_prompt(40, cfg["vocab_size"])
    outs = {}
    for kv in ("f32", "int8", "int4"):
        m = Model.synthetic(cfg, seed=0, max_seq_len=128, kv_dtype=kv)
        try:
            outs[kv] = m.forward_step(ids, 0).reshape(-1).copy()
        finally:
            m.close()
    assert rel(outs["int8"], outs["f32"]) < 2e-2
    assert rel(outs["int4"], outs["f32"]) < 0.3


@pytest.mark.parametrize("kv", ["int8", "int4"])
@pytest.mark.parametrize("mfma_min,wide_min", [(1, 8192), (160, 166)])
def test_quantised_kv_on_the_mfma_decode_kernel(kv, mfma_min, wide_min):
    """Long-context decode attention on the matrix cores over int8 / int4 pages: (code - offset) are exact bf16 integers
    fed straight to the MFMAs, the per-token scales multiply S^T rows (K) and p (V).  Forced from the first token and
    switched in mid-generation; single sequence and batched."""
    from crane_amd.backend import Model
    name = "tiny-qwen3-untied"
    cfg = configs.get_config(name)
    w = synth.synth_weights_f32(cfg, seed=0)
    o, o2 = _oracle(name, cfg, w, kv), _oracle(name, cfg, w, kv)
    tol = 2e-3 if kv == "int8" else 1e-3
    m = Model.synthetic(cfg, seed=0, max_seq_len=512, max_seqs=3, kv_dtype=kv)
    try:
        m.debug_set("attn_mfma_min", mfma_min)
        m.debug_set("attn_mfma_wide_min", wide_min)
        V = cfg["vocab_size"]
        ids = configs.synthetic_prompt(150, V)
        ref = o.forward(ids, 0)
        assert rel(m.forward_step(ids, 0).reshape(-1), ref) < tol
        tok = int(ref.argmax())
        for step in range(24):
            ref = o.forward([tok], 150 + step)
            got = m.forward_step([tok], 150 + step).reshape(-1)
            assert rel(got, ref) < tol, (step, rel(got, ref))
            tok = int(ref.argmax())
        s1, s2 = m.seq_alloc(), m.seq_alloc()
        m.seq_forward(s1, ids, 0, want_logits=False)
        m.seq_forward(s2, ids[:77], 0, want_logits=False)
        o.forward(ids, 0); o2.forward(ids[:77], 0)
        t1, t2 = 5, 9
        for step in range(8):
            lg, _ = m.step_batch_decode([s1, s2], [t1, t2])
            r1, r2 = o.forward([t1], 150 + step), o2.forward([t2], 77 + step)
            assert rel(lg[0].reshape(-1), r1) < tol and rel(lg[1].reshape(-1), r2) < tol, step
            t1, t2 = int(r1.argmax()), int(r2.argmax())
    finally:
        m.close()
