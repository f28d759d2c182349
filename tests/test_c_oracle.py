"""The C port (oracle/c) and the numpy oracle are independent restatements; they must agree."""
import os

import numpy as np
import pytest

from crane_amd import configs, synth
from oracle import c_oracle
from oracle.qwen3_oracle import Qwen3Config, Qwen3Oracle

pytestmark = pytest.mark.skipif(not os.path.exists(c_oracle.SO), reason="oracle/c not built (run __graft_entry__.build())")


@pytest.mark.parametrize("name", ["tiny-qwen3", "tiny-qwen3-untied"])
def test_c_port_matches_numpy_oracle(name):
    cfg = configs.get_config(name)
    w = synth.synth_weights_f32(cfg, 0)
    o = Qwen3Oracle(Qwen3Config.from_json(cfg), w)
    c = c_oracle.CQwen3(cfg, seed=0, max_seq=128)
    ids = configs.synthetic_prompt(11, cfg["vocab_size"])
    a, b = o.forward(ids, 0), c.forward(ids, 0)
    assert np.abs(a - b).max() / np.abs(a).max() < 2e-5
    a, b = o.forward([7], 11), c.forward([7], 11)
    assert np.abs(a - b).max() / np.abs(a).max() < 2e-5
    assert int(a.argmax()) == int(b.argmax())
    c.close()


def test_c_port_kv_rounding_mode():
    cfg = configs.get_config("tiny-qwen3")
    w = synth.synth_weights_f32(cfg, 0)
    o = Qwen3Oracle(Qwen3Config.from_json(cfg), w, kv_dtype="bf16")
    c = c_oracle.CQwen3(cfg, seed=0, max_seq=64, kv_bf16=True)
    ids = configs.synthetic_prompt(9, cfg["vocab_size"])
    a, b = o.forward(ids, 0), c.forward(ids, 0)
    # both sides round K/V to bf16 from values that differ in the last f32 bit, so a few elements
    # land on the other side of a bf16 tie (2^-9 on those elements, ~1e-4 on logits)
    assert np.abs(a - b).max() / np.abs(a).max() < 2e-4
    c.close()


def test_f16_rounding_point_is_ieee_binary16():
    """kv_bf16 = 2 rounds K/V like the device's f16 pages: RNE to binary16, subnormals kept, saturation at +-65504 -- pinned on
    numpy's float16 over normals, subnormals, ties and the overflow edge."""
    import ctypes
    lib = ctypes.CDLL(c_oracle.SO)
    lib.qc_f16_round.argtypes = [ctypes.c_float]
    lib.qc_f16_round.restype = ctypes.c_float
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.standard_normal(4000) * np.exp(rng.uniform(-20, 11, 4000)),
                         [0.0, -0.0, 1.0, 1.00048828125, 1.000244140625, 1.000732421875, 6.1035e-5, 6.0e-8, 2.98e-8, 2.99e-8, 5.97e-8,
                          65504.0, 65519.9, 65520.0, 1e6, -1e6, 3.14159e-7]]).astype(np.float32)
    want = np.clip(xs, -65504.0, 65504.0).astype(np.float16).astype(np.float32)
    got = np.array([lib.qc_f16_round(float(x)) for x in xs], dtype=np.float32)
    assert np.array_equal(got, want), xs[got != want][:8]
    from oracle.qwen3_oracle import f16_round
    assert np.array_equal(f16_round(xs), want)


@pytest.mark.parametrize("name,bar_f16,bf16_over", [("qwen3-8b-2l", 2e-4, True), ("qwen3-0.6b-2l", 1e-4, False)])
def test_kv_page_precision_against_the_hf_golden(name, bar_f16, bf16_over):
    """Why the device's default KV pages are binary16: on the HF f32 fixture at the headline widths a bf16 K/V append (the model
    dtype of the reference's GPU path) moves the logits by 1.06e-3 / 1.18e-3 (prompt / decode step) -- OUTSIDE north_star's
    1e-3 -- while a binary16 append, the same 2 bytes per element, measures 1.35e-4 / 1.32e-4."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"qwen3_{name}.npz"))
    cfg = configs.get_config(name)
    ids = g["prompt"].tolist()
    errs = {}
    for mode, label in ((2, "f16"), (1, "bf16")):
        c = c_oracle.CQwen3(cfg, seed=int(g["seed"][0]), max_seq=64, kv_bf16=mode)
        try:
            a = c.forward(ids, 0)
            b = c.forward(g["decode_token"].tolist(), len(ids))
            errs[label] = max(np.abs(a - g["prefill_logits"]).max() / np.abs(g["prefill_logits"]).max(),
                              np.abs(b - g["decode_logits"]).max() / np.abs(g["decode_logits"]).max())
        finally:
            c.close()
    assert errs["f16"] < bar_f16, errs
    assert errs["bf16"] > 5 * errs["f16"], errs
    if bf16_over:
        assert errs["bf16"] > 1e-3, errs


@pytest.mark.parametrize("fixture", ["tiny-qwen3.5", "qwen3.8-27b-geom4"])
def test_hybrid_c_port_matches_hf_golden(fixture):
    """oracle/c/qwen35_cpu.c (Gated Delta Net + gated attention, token-serial) is bench.py's CPU baseline and in-run parity checker
    for BASELINE configs[2].  Pinned here on HF Qwen3_5ForCausalLM (make_golden_qwen3_5.py): the tiny configuration and the REAL
    Qwen3.8-27B layer geometry (H 5120, n_rep 6, 3 value heads per key head, head_dim 256 with 64 rotary dims; 4 layers) --
    prompt logits, one decode step, the greedy continuation; and on the numpy oracle it is independent of."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"qwen3_5_{fixture}.npz"))
    if fixture == "tiny-qwen3.5":
        cfg = configs.get_config(fixture)
    else:
        cfg = dict(configs.get_config("qwen3.8-27b"), num_hidden_layers=4, vocab_size=4096, max_position_embeddings=4096)
    c = c_oracle.CQwen35(cfg, seed=int(g["seed"][0]), max_seq=128)
    try:
        ids = g["prompt"].tolist()
        a = c.forward(ids, 0)
        assert np.abs(a - g["prefill_logits"]).max() / np.abs(g["prefill_logits"]).max() < 2e-5
        b = c.forward(g["decode_token"].tolist(), len(ids))
        assert np.abs(b - g["decode_logits"]).max() / np.abs(g["decode_logits"]).max() < 2e-5
        want = g["greedy_tokens"].tolist()
        lg, toks = c.forward(ids, 0), list(ids)
        for _ in range(len(want) - len(ids)):
            toks.append(int(lg.argmax()))
            lg = c.forward([toks[-1]], len(toks) - 1)
        assert toks == want
        if fixture == "tiny-qwen3.5":
            from oracle import qwen3_5_oracle as O5
            w = synth.synth_weights_f32(cfg, int(g["seed"][0]))
            for kv, mode in (("f32", 0), ("f16", 2)):
                o = O5.Qwen35Oracle(O5.Qwen35Config.from_json(cfg), w, kv_dtype=kv)
                c2 = c_oracle.CQwen35(cfg, seed=int(g["seed"][0]), max_seq=128, kv_round=mode)
                p = configs.synthetic_prompt(70, cfg["vocab_size"])
                x, y = o.forward(p, 0), c2.forward(p, 0)
                assert np.abs(x - y).max() / np.abs(x).max() < (2e-5 if mode == 0 else 1e-4)
                c2.close()
    finally:
        c.close()


@pytest.mark.parametrize("name", ["qwen3-8b-2l", "qwen3-0.6b-2l"])
def test_c_port_matches_hf_golden_at_the_headline_geometry(name):
    """oracle/c is the checker of the headline parity tests (tests/test_gpu_parity_headline.py) and of bench.py's parity leg.
    Here it is pinned itself, at those geometries: Qwen3-8B widths, GQA 4, the 151 936-row untied lm_head (BASELINE
    configs[1]) and Qwen3-0.6B widths with the tied table (configs[0]), 2 layers each, against HF Qwen3ForCausalLM on the
    same synthetic checkpoint (tests/golden/make_golden_qwen3.py): prompt logits, one decode step, 12 greedy tokens."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"qwen3_{name}.npz"))
    cfg = configs.get_config(name)
    c = c_oracle.CQwen3(cfg, seed=int(g["seed"][0]), max_seq=64)
    try:
        ids = g["prompt"].tolist()
        a = c.forward(ids, 0)
        assert np.abs(a - g["prefill_logits"]).max() / np.abs(g["prefill_logits"]).max() < 2e-5
        b = c.forward(g["decode_token"].tolist(), len(ids))
        assert np.abs(b - g["decode_logits"]).max() / np.abs(g["decode_logits"]).max() < 2e-5
        want = g["greedy_tokens"].tolist()
        lg, toks = c.forward(ids, 0), list(ids)
        for _ in range(len(want) - len(ids)):
            toks.append(int(lg.argmax()))
            lg = c.forward([toks[-1]], len(toks) - 1)
        assert toks == want
    finally:
        c.close()


def test_c_port_runs_the_text_model_of_a_vl_checkpoint():
    """bench.py --model qwen3-vl-2b checks its decode against the C port too: the dense decoder of a Qwen3-VL checkpoint keeps its
    tensors under model.language_model. (qc_set_name_prefix) and rotates text rows like plain RoPE (T = H = W): the port equals
    the numpy Qwen3-VL oracle on a text-only prompt and a decode step."""
    from oracle.qwen3_vl_oracle import Qwen3VLOracle
    cfg = configs.get_config("tiny-qwen3-vl")
    c = c_oracle.CQwen3(cfg, seed=0, max_seq=64)
    try:
        ids = configs.synthetic_prompt(9, 400)
        a = c.forward(ids, 0)
        o = Qwen3VLOracle(cfg, synth.synth_weights_f32(cfg, 0))
        _, _, b = o.prefill(ids, np.zeros((0, 1536), np.float32), [])
        assert np.abs(a - b).max() / np.abs(a).max() < 2e-5
        a2, b2 = c.forward([int(a.argmax())], 9), np.asarray(o.decode(int(b.argmax()))).reshape(-1)
        assert np.abs(a2 - b2).max() / np.abs(a2).max() < 2e-5
    finally:
        c.close()
        c_oracle.CQwen3(configs.get_config("tiny-qwen3"), seed=0, max_seq=16).close()      # (resets the name prefix)


@pytest.mark.parametrize("name,n", [("tiny-qwen3", 37), ("tiny-qwen3-untied", 6)])
def test_batched_prompt_pass_is_bit_identical(name, n):
    """qc_forward_batched (every weight matrix streamed once for the whole prompt: what makes a 1024-token, 36-layer reference
    affordable in bench.py's parity leg) is qc_forward's arithmetic position by position -- logits AND the cache it leaves."""
    cfg = configs.get_config(name)
    a = c_oracle.CQwen3(cfg, seed=0, max_seq=96)
    b = c_oracle.CQwen3(cfg, seed=0, max_seq=96)
    ids = configs.synthetic_prompt(n, cfg["vocab_size"])
    assert np.array_equal(a.forward(ids, 0), b.forward_batched(ids, 0))
    assert np.array_equal(a.forward([5], n), b.forward([5], n))                       # decode over the cache each pass wrote
    more = configs.synthetic_prompt(9, cfg["vocab_size"])
    assert np.array_equal(a.forward(more, n + 1), b.forward_batched(more, n + 1))     # a second chunk on a non-empty cache
    a.close(); b.close()
