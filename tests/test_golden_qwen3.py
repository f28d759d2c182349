"""Golden vectors produced by HuggingFace transformers (tests/golden/make_golden_qwen3.py).

CPU: pins the numpy oracle and the C port against an implementation we did not write.
GPU: pins the HIP path (through the C ABI) against the same vectors.
"""
import os

import numpy as np
import pytest

from crane_amd import configs, synth
from oracle.qwen3_oracle import Qwen3Config, Qwen3Oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = ["tiny-qwen3", "tiny-qwen3-untied"]


def _load(name):
    g = np.load(os.path.join(GOLD, f"qwen3_{name}.npz"))
    cfg = configs.get_config(name)
    return g, cfg, synth.synth_weights_f32(cfg, seed=int(g["seed"][0]))


def rel(a, ref):
    return float(np.abs(a - ref).max() / np.abs(ref).max())


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_hf_golden(name):
    g, cfg, w = _load(name)
    o = Qwen3Oracle(Qwen3Config.from_json(cfg), w)
    ids = g["prompt"].tolist()
    assert rel(o.forward(ids, 0), g["prefill_logits"]) < 2e-5
    assert rel(o.forward(g["decode_token"].tolist(), len(ids)), g["decode_logits"]) < 2e-5
    n_new = len(g["greedy_tokens"]) - len(ids)
    assert o.generate(ids, n_new) == g["greedy_tokens"].tolist()


@pytest.mark.parametrize("name", NAMES)
def test_c_port_matches_hf_golden(name):
    from oracle import c_oracle
    if not os.path.exists(c_oracle.SO):
        pytest.skip("oracle/c not built")
    g, cfg, _ = _load(name)
    c = c_oracle.CQwen3(cfg, seed=int(g["seed"][0]), max_seq=64)
    ids = g["prompt"].tolist()
    assert rel(c.forward(ids, 0), g["prefill_logits"]) < 2e-5
    assert rel(c.forward(g["decode_token"].tolist(), len(ids)), g["decode_logits"]) < 2e-5
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("kv", ["f32", "f16", "bf16"])
def test_hip_path_matches_hf_golden(name, kv):
    from crane_amd.backend import GenerationConfig, Model
    g, cfg, _ = _load(name)
    m = Model.synthetic(cfg, seed=int(g["seed"][0]), max_seq_len=128, max_seqs=2, kv_dtype=kv)
    try:
        ids = g["prompt"].tolist()
        # vs the HF f32 forward.  f32 pages: 1e-4; f16 pages (default, benchmarked): the 1e-3 north-star bar; bf16 pages
        # (opt-in): one bf16 epsilon, not claimed to meet the bar
        tol = {"f32": 1e-4, "f16": 1e-3, "bf16": 4e-3}[kv]
        assert rel(m.forward_step(ids, 0)[0, 0], g["prefill_logits"]) < tol
        assert rel(m.forward_step(g["decode_token"].tolist(), len(ids))[0, 0], g["decode_logits"]) < tol
        n_new = len(g["greedy_tokens"]) - len(ids)
        assert m.generate(ids, GenerationConfig.greedy(n_new)) == g["greedy_tokens"].tolist()   # bit-exact ids
    finally:
        m.close()
