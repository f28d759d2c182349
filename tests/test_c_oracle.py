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
