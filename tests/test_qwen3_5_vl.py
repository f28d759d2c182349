"""Qwen 3.5-VL: vision tower + VLM glue.  CPU: oracle vs HF golden and the reference's position / layout KATs.
GPU: HIP path (C ABI) vs oracle (reference GELU form) and vs the HF golden (erf form)."""
import os

import numpy as np
import pytest

from crane_amd import configs, synth
from oracle import qwen3_5_vision_oracle as VO
from oracle.qwen3_5_oracle import Qwen35Config, Qwen35Oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "qwen3_5_vl_tiny.npz")


def rel(a, ref):
    return float(np.abs(a - ref).max() / np.abs(ref).max())


def _setup():
    g = np.load(GOLD)
    cfg = configs.get_config("tiny-qwen3.5-vl")
    w = synth.synth_weights_f32(cfg, int(g["seed"][0]))
    text_w = {k.replace("model.language_model.", "model."): v for k, v in w.items() if not k.startswith("model.visual.")}
    return g, cfg, w, text_w


def _oracle_vlm(cfg, w, text_w, ids, pix, grid, merger_gelu, n_new=0):
    vo = VO.VisionOracle(cfg["vision_config"], w, merger_gelu=merger_gelu)
    feat = vo.forward(pix, grid)
    IMG = cfg["image_token_id"]
    pos3, nxt = VO.build_position_ids(ids, grid, IMG, cfg["vision_config"]["spatial_merge_size"])
    o = Qwen35Oracle(Qwen35Config.from_json(cfg), text_w)
    emb = VO.splice_image_features(ids, o.embed[np.array(ids)], feat, IMG)
    logits = o.forward(ids, 0, embeds=emb, pos3=pos3)
    toks, lg, p = [], logits, nxt
    for i in range(n_new):
        t = int(np.argmax(lg)); toks.append(t)
        lg = o.forward([t], len(ids) + i, pos3=np.array([[p]] * 3)); p += 1
    return feat, logits, toks


def test_position_ids_kat():
    """vlm.rs:190-241: text tokens advance all three axes; an image span of (t,h,w)=(1,2,3) merged tokens gets
    (base, base + row, base + col) and the next text position is base + max(t, h, w)."""
    IMG = 9
    pos, nxt = VO.build_position_ids([1, 2, IMG, IMG, IMG, IMG, IMG, IMG, 3], [[1, 4, 6]], IMG, 2)
    assert pos[:, :2].tolist() == [[0, 1]] * 3
    assert pos[0, 2:8].tolist() == [2] * 6
    assert pos[1, 2:8].tolist() == [2, 2, 2, 3, 3, 3] and pos[2, 2:8].tolist() == [2, 3, 4, 2, 3, 4]
    assert pos[:, 8].tolist() == [5, 5, 5] and nxt == 6


def test_block_major_patch_order_kat():
    """processor.rs:282-329: for a 2x4-patch image the raster patch ids appear in merge-block-major order
    [0,1,4,5,2,3,6,7]; rot_pos_emb / pos-embed use the same order (vision.rs:468-489,502-526)."""
    h, w, m = 2, 4, 2
    order = [(br * m + ir) * w + bc * m + ic for br in range(h // m) for bc in range(w // m) for ir in range(m) for ic in range(m)]
    assert order == [0, 1, 4, 5, 2, 3, 6, 7]
    cfg = configs.get_config("tiny-qwen3.5-vl")
    vo = VO.VisionOracle(cfg["vision_config"], synth.synth_weights_f32(cfg, 0))
    rot = vo.rot_pos_emb([[1, 2, 4]])
    quarter = rot.shape[1] // 2
    rows = np.round(rot[:, 0] / rot[1, 0] if rot[1, 0] else rot[:, 0])       # column 0 carries row * inv_freq[0] (= row)
    assert rot[:, 0].tolist() == [0, 0, 1, 1, 0, 0, 1, 1] and rot[:, quarter].tolist() == [0, 1, 0, 1, 2, 3, 2, 3]


def test_vision_shape_kat():
    """crane-core/tests/qwen3_5_vision.rs:56-85: N patches in -> N/4 merged tokens of out_hidden width."""
    cfg = configs.get_config("tiny-qwen3.5-vl")
    vo = VO.VisionOracle(cfg["vision_config"], synth.synth_weights_f32(cfg, 0))
    pix = np.zeros((6 * 6, 1536), np.float32)
    assert vo.forward(pix, [[1, 6, 6]]).shape == (9, cfg["vision_config"]["out_hidden_size"])


def test_oracle_matches_hf_golden():
    g, cfg, w, text_w = _setup()
    feat, logits, toks = _oracle_vlm(cfg, w, text_w, g["input_ids"].tolist(), g["pixel_values"], g["grid_thw"].tolist(), "erf", 6)
    assert rel(feat, g["features"]) < 2e-5 and rel(logits, g["prefill_logits"]) < 2e-5
    assert toks == g["greedy_tokens"].tolist()[len(g["input_ids"]):]


def test_oracle_matches_hf_golden_at_the_real_size():
    """The reference's LIVE image path at real size: the 24 x 1024 tower of qwen3.5-vl-0.8b in front of the Qwen3.5-0.8B text
    geometry (H 1024, 8 q / 2 kv heads of 256 with 64 rotary dims and 3-axis MRoPE, 16 GDN heads of 128, I 3584, the
    248 320-entry tied table; 4 layers = 3 GDN + 1 gated attention).  96 patches -> 24 merged tokens; golden from
    tests/golden/make_golden_qwen3_5_vl.py tower24."""
    g = np.load(os.path.join(os.path.dirname(GOLD), "qwen3_5_vl_tower24.npz"))
    cfg = configs.get_config("qwen3.5-vl-0.8b")
    cfg = dict(cfg, text_config=dict(cfg["text_config"], num_hidden_layers=4, max_position_embeddings=4096))
    w = synth.synth_weights_f32(cfg, int(g["seed"][0]))
    text_w = {k.replace("model.language_model.", "model."): v for k, v in w.items() if not k.startswith("model.visual.")}
    grid = g["grid_thw"].tolist()
    pix = np.random.default_rng(0).standard_normal((grid[0][1] * grid[0][2], 3 * 2 * 16 * 16)).astype(np.float32)
    ids = g["input_ids"].tolist()
    feat, logits, toks = _oracle_vlm(cfg, w, text_w, ids, pix, grid, "erf", 6)
    assert feat.shape == (24, 1024) and rel(feat, g["features"]) < 2e-5
    assert rel(logits, g["prefill_logits"]) < 5e-5
    assert toks == g["greedy_tokens"].tolist()[len(ids):]


@pytest.mark.gpu
@pytest.mark.parametrize("gelu", ["tanh", "erf"])
def test_hip_vision_and_vlm(gelu):
    from crane_amd.backend import Model
    g, cfg, w, text_w = _setup()
    ids, pix, grid = g["input_ids"].tolist(), g["pixel_values"], g["grid_thw"].tolist()
    m = Model.synthetic(cfg, seed=0, max_seq_len=256, max_seqs=2, kv_dtype="f32")
    try:
        m.debug_set("vision_merger_gelu", 2 if gelu == "erf" else 1)      # PatchMerger GELU: tanh form (reference) / erf (HF)
        assert m.image_token_id() == cfg["image_token_id"]
        feat_ref, logits_ref, toks_ref = _oracle_vlm(cfg, w, text_w, ids, pix, grid, gelu, 6)
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
        # a text-only prompt on the same checkpoint still works (positions T == H == W)
        m.clear_kv_cache()
        o = Qwen35Oracle(Qwen35Config.from_json(cfg), text_w)
        t_ids = configs.synthetic_prompt(11, 400)
        assert rel(m.forward_step(t_ids, 0)[0, 0], o.forward(t_ids, 0)) < 1e-4
    finally:
        m.close()


@pytest.mark.gpu
def test_forward_embeds_equals_seq_forward_and_vlm_forward():
    """cm_embed_tokens / cm_forward_embeds = Qwen3_5TextModel::embed_only / forward_embeds (qwen3_5/model.rs:368,430): text-only
    rows at 1-D positions must reproduce cm_seq_forward bit for bit (same kernels, the rows are exactly the table's), and the
    image prompt assembled BY THE CALLER -- text embeddings, image features from cm_vision_encode spliced over the placeholder
    rows (vlm.rs:433-468), 3-axis positions (vlm.rs:190-241) -- must reproduce cm_vlm_forward, decode steps included."""
    from crane_amd.backend import Model
    g, cfg, w, text_w = _setup()
    ids, pix, grid = g["input_ids"].tolist(), g["pixel_values"], g["grid_thw"].tolist()
    m = Model.synthetic(cfg, seed=0, max_seq_len=256, max_seqs=4, kv_dtype="f32")
    try:
        t_ids = configs.synthetic_prompt(23, 400)
        emb = m.embed_tokens(t_ids)
        assert np.array_equal(emb, text_w["model.embed_tokens.weight"][np.asarray(t_ids)])
        a, ga = m.seq_forward(0, t_ids, 0)
        s1 = m.seq_alloc()
        pos = np.tile(np.arange(len(t_ids), dtype=np.int32), (3, 1))
        b, gb = m.forward_embeds(emb, pos, 0, seq=s1)
        assert np.array_equal(a, b) and ga == gb
        s2 = m.seq_alloc()
        c, gc = m.forward_embeds(emb, None, 0, seq=s2)               # no positions: the sequence's own counter
        assert np.array_equal(a, c) and ga == gc
        a2, _ = m.seq_forward(0, [7], len(t_ids)); b2, _ = m.seq_forward(s1, [7], len(t_ids))
        assert np.array_equal(a2, b2)                                # decode continues identically
        # image prompt
        m.clear_kv_cache(); m.seq_free(s1); m.seq_free(s2)
        ref, nxt = m.vlm_forward(ids, pix, grid)
        ref2 = m.forward_step([nxt], len(ids))[0, 0]
        feat = m.encode_images(pix, grid)
        e = m.embed_tokens(ids)
        img_tok = cfg["image_token_id"]
        rows = [i for i, t in enumerate(ids) if t == img_tok]
        assert len(rows) == feat.shape[0]
        e[rows] = feat
        merge = cfg["vision_config"]["spatial_merge_size"]
        s3 = m.seq_alloc()
        pos3, _ = VO.build_position_ids(ids, grid, img_tok, merge)   # build_position_ids (qwen3_5/vlm.rs:190-241) as the oracle restates it
        got, g3 = m.forward_embeds(e, np.asarray(pos3, dtype=np.int32).reshape(3, len(ids)), 0, seq=s3)
        assert rel(got, ref) < 1e-6 and g3 == nxt, rel(got, ref)
        got2, _ = m.seq_forward(s3, [nxt], len(ids))                 # rotary position = cache position + MRoPE delta
        assert rel(got2, ref2) < 1e-6, rel(got2, ref2)
    finally:
        m.close()


@pytest.mark.gpu
def test_hip_live_image_path_at_the_real_size_against_the_hf_golden():
    """The reference's LIVE image path at real size on the GPU: the 24 x 1024 tower (16 heads of 64, 2304 interpolated position
    embeddings, merger to 1024) in front of the Qwen3.5-0.8B text geometry (3 GDN + 1 gated-attention layer, 248 320-entry tied
    table, 3-axis MRoPE) against HF Qwen3_5ForConditionalGeneration on the committed fixture
    tests/golden/qwen3_5_vl_tower24.npz (make_golden_qwen3_5_vl.py tower24): tower features, image+text prompt logits, the
    greedy continuation -- default KV pages (f16), bar 1e-3 (features: 1e-4, no cache involved)."""
    from crane_amd.backend import Model
    g = np.load(os.path.join(os.path.dirname(GOLD), "qwen3_5_vl_tower24.npz"))
    cfg = configs.get_config("qwen3.5-vl-0.8b")
    cfg = dict(cfg, text_config=dict(cfg["text_config"], num_hidden_layers=4, max_position_embeddings=4096))
    grid = g["grid_thw"].tolist()
    pix = np.random.default_rng(0).standard_normal((grid[0][1] * grid[0][2], 3 * 2 * 16 * 16)).astype(np.float32)
    ids = g["input_ids"].tolist()
    m = Model.synthetic(cfg, seed=int(g["seed"][0]), max_seq_len=256, max_seqs=2)
    try:
        m.debug_set("vision_merger_gelu", 2)                         # HF's erf form
        feat = m.encode_images(pix, grid)
        assert feat.shape == (24, 1024) and rel(feat, g["features"]) < 1e-4, rel(feat, g["features"])
        logits, nxt = m.vlm_forward(ids, pix, grid)
        assert rel(logits, g["prefill_logits"]) < 1e-3, rel(logits, g["prefill_logits"])
        toks, pos = [nxt], len(ids)
        want = g["greedy_tokens"].tolist()[len(ids):]
        for _ in range(len(want) - 1):
            toks.append(m.forward_step_greedy([toks[-1]], pos)); pos += 1
        assert toks == want
    finally:
        m.close()


@pytest.mark.gpu
def test_hip_vlm_two_images_of_different_grids():
    """The live image path with n_images = 2 (24- and 16-patch frames) and text before, between and after the images: tower
    features per frame, splice over both placeholder spans, MRoPE positions across two images, decode on the counter."""
    from crane_amd.backend import Model
    g, cfg, w, text_w = _setup()
    rng = np.random.default_rng(5)
    grid = [[1, 4, 6], [1, 2, 8]]
    pix = rng.standard_normal((24 + 16, 3 * 2 * 16 * 16)).astype(np.float32)
    img, vs, ve = cfg["image_token_id"], cfg["vision_start_token_id"], cfg["vision_end_token_id"]
    ids = [5, 6, vs] + [img] * 6 + [ve, 7, 8, 9, vs] + [img] * 4 + [ve, 10, 11]
    m = Model.synthetic(cfg, seed=0, max_seq_len=256, max_seqs=2, kv_dtype="f32")
    try:
        feat_ref, logits_ref, toks_ref = _oracle_vlm(cfg, w, text_w, ids, pix, grid, "tanh", 6)
        feat = m.encode_images(pix, grid)
        assert feat.shape == feat_ref.shape and feat.shape[0] == 10 and rel(feat, feat_ref) < 1e-4
        logits, nxt = m.vlm_forward(ids, pix, grid)
        assert rel(logits, logits_ref) < 1e-4 and nxt == toks_ref[0]
        toks, pos = [nxt], len(ids)
        for _ in range(5):
            toks.append(m.forward_step_greedy([toks[-1]], pos)); pos += 1
        assert toks == toks_ref
    finally:
        m.close()
