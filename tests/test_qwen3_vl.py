"""Qwen3-VL (BASELINE configs[3]): vision tower with DeepStack mergers + dense Qwen3 decoder with 3-axis MRoPE.
CPU: the oracle (oracle/qwen3_vl_oracle.py) against the HF golden (tests/golden/make_golden_qwen3_vl.py).
GPU: the HIP path through the C ABI against the oracle (reference GELU form) and the HF golden (erf form)."""
import os

import numpy as np
import pytest

from crane_amd import configs, synth
from oracle.qwen3_vl_oracle import Qwen3VLOracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "qwen3_vl_tiny.npz")


def rel(a, ref):
    return float(np.abs(a - ref).max() / np.abs(ref).max())


def _setup():
    g = np.load(GOLD)
    cfg = configs.get_config("tiny-qwen3-vl")
    return g, cfg, synth.synth_weights_f32(cfg, int(g["seed"][0]))


def _oracle_run(cfg, w, ids, pix, grid, gelu, n_new):
    o = Qwen3VLOracle(cfg, w, merger_gelu=gelu)
    feat, deep, logits = o.prefill(ids, pix, grid)
    toks, lg = [], logits
    for _ in range(n_new):
        t = int(np.argmax(lg)); toks.append(t)
        lg = o.decode(t)
    return feat, deep, logits, toks


def test_oracle_matches_hf_golden():
    g, cfg, w = _setup()
    ids = g["input_ids"].tolist()
    feat, deep, logits, toks = _oracle_run(cfg, w, ids, g["pixel_values"], g["grid_thw"].tolist(), "erf", 6)
    assert rel(feat, g["features"]) < 2e-5
    assert len(deep) == g["deepstack"].shape[0] == 2
    for k in range(2):
        assert rel(deep[k], g["deepstack"][k]) < 2e-5, k
    assert rel(logits, g["prefill_logits"]) < 5e-5
    assert toks == g["greedy_tokens"].tolist()[len(ids):]


def test_oracle_matches_hf_golden_at_the_real_tower_size():
    """The same at the REAL Qwen3-VL-2B vision tower (depth 24, hidden 1024, 16 heads of 64, 2304 interpolated position
    embeddings, DeepStack after blocks 5 / 11 / 17, merger to 2048) and text widths (4 of the 28 decoder layers, 151 936-entry
    tied table): 96 patches -> 24 merged tokens, golden from tests/golden/make_golden_qwen3_vl.py tower24.  (~1 min of
    synthetic-weight generation.)"""
    g = np.load(os.path.join(os.path.dirname(GOLD), "qwen3_vl_tower24.npz"))
    cfg = configs.get_config("qwen3-vl-2b")
    cfg = dict(cfg, text_config=dict(cfg["text_config"], num_hidden_layers=4, max_position_embeddings=4096))
    w = synth.synth_weights_f32(cfg, int(g["seed"][0]))
    grid = g["grid_thw"].tolist()
    pix = np.random.default_rng(0).standard_normal((grid[0][1] * grid[0][2], 3 * 2 * 16 * 16)).astype(np.float32)
    ids = g["input_ids"].tolist()
    feat, deep, logits, toks = _oracle_run(cfg, w, ids, pix, grid, "erf", 6)
    assert feat.shape == (24, 2048) and rel(feat, g["features"]) < 2e-5
    assert len(deep) == g["deepstack"].shape[0] == 3
    for k in range(3):
        assert rel(deep[k], g["deepstack"][k]) < 2e-5, k
    assert rel(logits, g["prefill_logits"]) < 5e-5
    assert toks == g["greedy_tokens"].tolist()[len(ids):]


def test_text_only_prompt_equals_dense_qwen3():
    """without images T = H = W: the MRoPE rows are the plain RoPE rows and nothing is injected"""
    from oracle.qwen3_oracle import Qwen3Config, Qwen3Oracle
    g, cfg, w = _setup()
    o = Qwen3VLOracle(cfg, w)
    ids = configs.synthetic_prompt(9, 400)
    _, _, a = o.prefill(ids, np.zeros((0, 1536), np.float32), [])
    t = dict(cfg["text_config"], model_type="qwen3", tie_word_embeddings=cfg["tie_word_embeddings"], rope_theta=5e6)
    text_w = {k.replace("model.language_model.", "model."): v for k, v in w.items() if not k.startswith("model.visual.")}
    b = Qwen3Oracle(Qwen3Config.from_json(t), text_w).forward(ids, 0)
    assert rel(a, b) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("gelu", ["tanh", "erf"])
def test_hip_qwen3_vl(gelu):
    from crane_amd.backend import Model
    g, cfg, w = _setup()
    ids, pix, grid = g["input_ids"].tolist(), g["pixel_values"], g["grid_thw"].tolist()
    m = Model.synthetic(cfg, seed=0, max_seq_len=256, max_seqs=2, kv_dtype="f32")
    try:
        m.debug_set("vision_merger_gelu", 2 if gelu == "erf" else 1)      # PatchMerger GELU: tanh form (reference) / erf (HF)
        assert m.image_token_id() == cfg["image_token_id"]
        feat_ref, deep_ref, logits_ref, toks_ref = _oracle_run(cfg, w, ids, pix, grid, gelu, 6)
        feat = m.encode_images(pix, grid)
        assert feat.shape == feat_ref.shape and rel(feat, feat_ref) < 1e-4
        logits, nxt = m.vlm_forward(ids, pix, grid)
        assert rel(logits, logits_ref) < 1e-4 and nxt == toks_ref[0]
        toks, pos = [nxt], len(ids)
        for _ in range(5):                                  # decode continues with the MRoPE counter
            toks.append(m.forward_step_greedy([toks[-1]], pos)); pos += 1
        assert toks == toks_ref
        if gelu == "erf":                                   # independent implementation (HF)
            assert rel(feat, g["features"]) < 1e-4 and rel(logits, g["prefill_logits"]) < 1e-4
            assert toks == g["greedy_tokens"].tolist()[len(ids):]
    finally:
        m.close()


@pytest.mark.gpu
def test_hip_qwen3_vl_at_the_real_tower_size_against_the_hf_golden():
    """BASELINE configs[3] on the GPU at the REAL Qwen3-VL-2B vision tower (depth 24, hidden 1024, 16 heads of 64, DeepStack taps
    after blocks 5 / 11 / 17, merger to 2048) and text widths (2048 / 16 q / 8 kv heads -- GQA group 2 -- 4 of the 28 layers, the
    151 936-entry tied table) against HF Qwen3VLForConditionalGeneration on the committed fixture
    tests/golden/qwen3_vl_tower24.npz (make_golden_qwen3_vl.py tower24): tower features, the three DeepStack feature maps,
    image+text prompt logits with the DeepStack injection after decoder layers 0-2 (qwen3_vl/text.rs:280-333), and the greedy
    continuation -- default KV pages (f16), bar 1e-3 on logits, 1e-4 on the tower outputs."""
    from crane_amd.backend import Model
    g = np.load(os.path.join(os.path.dirname(GOLD), "qwen3_vl_tower24.npz"))
    cfg = configs.get_config("qwen3-vl-2b")
    cfg = dict(cfg, text_config=dict(cfg["text_config"], num_hidden_layers=4, max_position_embeddings=4096))
    grid = g["grid_thw"].tolist()
    pix = np.random.default_rng(0).standard_normal((grid[0][1] * grid[0][2], 3 * 2 * 16 * 16)).astype(np.float32)
    ids = g["input_ids"].tolist()
    m = Model.synthetic(cfg, seed=int(g["seed"][0]), max_seq_len=256, max_seqs=2)
    try:
        m.debug_set("vision_merger_gelu", 2)                         # HF's erf form
        feat = m.encode_images(pix, grid)
        assert feat.shape == (24, 2048) and rel(feat, g["features"]) < 1e-4, rel(feat, g["features"])
        deep = m.debug_read("deepstack", 3 * 24 * 2048).reshape(3, 24, 2048)
        for k in range(3):
            assert rel(deep[k], g["deepstack"][k]) < 1e-4, (k, rel(deep[k], g["deepstack"][k]))
        logits, nxt = m.vlm_forward(ids, pix, grid)
        assert rel(logits, g["prefill_logits"]) < 1e-3, rel(logits, g["prefill_logits"])
        toks, pos = [nxt], len(ids)
        want = g["greedy_tokens"].tolist()[len(ids):]
        for _ in range(len(want) - 1):
            toks.append(m.forward_step_greedy([toks[-1]], pos)); pos += 1
        assert toks == want
    finally:
        m.close()


def _two_images(cfg, seed=5):
    """two images of different grids with text before, between and after them"""
    rng = np.random.default_rng(seed)
    grid = [[1, 4, 6], [1, 2, 8]]                      # 24 + 16 patches -> 6 + 4 merged tokens
    pix = rng.standard_normal((24 + 16, 3 * 2 * 16 * 16)).astype(np.float32)
    img, vs, ve = cfg["image_token_id"], cfg["vision_start_token_id"], cfg["vision_end_token_id"]
    ids = [5, 6, vs] + [img] * 6 + [ve, 7, 8, 9, vs] + [img] * 4 + [ve, 10, 11]
    return ids, pix, grid


@pytest.mark.gpu
def test_hip_qwen3_vl_two_images_of_different_grids():
    """cm_vision_encode / cm_vlm_forward with n_images = 2 (frames of 24 and 16 patches: per-frame bidirectional attention,
    per-image position-embedding interpolation), text between the images (3-axis MRoPE positions restart from the running
    maximum after each image, build_position_ids qwen3_5/vlm.rs:190-241), DeepStack rows of both images added after the
    first decoder layers -- against the oracle, then 5 decode steps on the MRoPE counter."""
    from crane_amd.backend import Model
    g, cfg, w = _setup()
    ids, pix, grid = _two_images(cfg)
    m = Model.synthetic(cfg, seed=0, max_seq_len=256, max_seqs=2, kv_dtype="f32")
    try:
        feat_ref, deep_ref, logits_ref, toks_ref = _oracle_run(cfg, w, ids, pix, grid, "tanh", 6)
        feat = m.encode_images(pix, grid)
        assert feat.shape == feat_ref.shape == (10, cfg["text_config"]["hidden_size"]) and rel(feat, feat_ref) < 1e-4
        logits, nxt = m.vlm_forward(ids, pix, grid)
        assert rel(logits, logits_ref) < 1e-4 and nxt == toks_ref[0]
        toks, pos = [nxt], len(ids)
        for _ in range(5):
            toks.append(m.forward_step_greedy([toks[-1]], pos)); pos += 1
        assert toks == toks_ref
    finally:
        m.close()
