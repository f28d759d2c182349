"""ONE handle that owns all tensor-parallel ranks (cm_opts.tp_mode = CM_TP_IN_PROCESS), run with every rank on ONE GPU (the
1-GPU test mode: tp_devices = [0] * tp): the C++ loader's shards of ALL ranks, the library's fan-out of every cm_* call to
the rank threads, and the real exchange steps -- the peer-store all-reduce behind o_proj / out_proj and down_proj, the
all-gather of the arg-max partials / logits shards (csrc/kernels_tp.hip; RCCL refuses two ranks on one device) -- must
reproduce the unsharded model: logits against the f32 CPU oracle (north_star's 1e-3 bar) and against the TP = 1 handle
(summation order only), greedy ids equal.  What the gloo tests prove about the plan and test_gpu_tp_shards.py about one rank
at a time, this proves end to end through the ABI a crane-serve ModelBackend would hold."""
import os

import numpy as np
import pytest

from crane_amd import configs, synth

pytestmark = pytest.mark.gpu
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")      # one hardware queue per rank stream when the ranks share a device


def rel(a, ref):
    return float(np.abs(a - ref).max() / np.abs(ref).max())


def _group(cfg, tp, **kw):
    from crane_amd.backend import Model
    kw.setdefault("max_seq_len", 256)
    kw.setdefault("max_seqs", 4)
    return Model.synthetic(cfg, seed=0, tp_size=tp, tp_in_process=True, tp_devices=[0] * tp, **kw)


def _single(cfg, **kw):
    from crane_amd.backend import Model
    kw.setdefault("max_seq_len", 256)
    kw.setdefault("max_seqs", 4)
    return Model.synthetic(cfg, seed=0, **kw)


@pytest.mark.parametrize("tp", [2, 4])
def test_dense_group_equals_the_unsharded_model(tp):
    from crane_amd.backend import GenerationConfig
    from oracle.qwen3_oracle import Qwen3Config, Qwen3Oracle
    cfg = configs.get_config("tiny-qwen3-untied")                     # 8 q heads, 2 kv heads (replicated at tp = 4), V = 1000
    w = synth.synth_weights_f32(cfg, 0)
    o = Qwen3Oracle(Qwen3Config.from_json(cfg), w)
    ids = configs.synthetic_prompt(21, cfg["vocab_size"])
    g, s = _group(cfg, tp), _single(cfg)
    try:
        assert g.tp_ranks() == tp
        a, b = g.forward_step(ids, 0)[0, 0], s.forward_step(ids, 0)[0, 0]      # prompt pass: [S, H] all-reduces, logits gather
        assert rel(a, o.forward(ids, 0)) < 1e-3 and rel(a, b) < 1e-4
        a, b = g.forward_step([5], len(ids))[0, 0], s.forward_step([5], len(ids))[0, 0]   # decode step: [H] all-reduces
        assert rel(a, o.forward([5], len(ids))) < 1e-3 and rel(a, b) < 1e-4
        # greedy chain: hipGraph replays (the exchange kernels' epochs advance inside the graph), arg-max partials gathered
        ta = g.generate(ids, GenerationConfig.greedy(24))
        tb = s.generate(ids, GenerationConfig.greedy(24))
        assert ta == tb
        # sampled tokens: every rank draws from the same gathered logits with the same counter-based stream
        gc = GenerationConfig(max_new_tokens=12, temperature=0.8, top_p=0.9, seed=7)
        assert g.generate(ids, gc) == s.generate(ids, gc)
    finally:
        g.close(); s.close()


def test_hybrid_group_equals_the_unsharded_model():
    from crane_amd.backend import GenerationConfig
    cfg = configs.get_config("tiny-qwen3.5")
    ids = configs.synthetic_prompt(21, cfg["vocab_size"])
    g, s = _group(cfg, 2), _single(cfg)
    try:
        a, b = g.forward_step(ids, 0)[0, 0], s.forward_step(ids, 0)[0, 0]
        assert rel(a, b) < 1e-4
        a, b = g.forward_step([5], len(ids))[0, 0], s.forward_step([5], len(ids))[0, 0]
        assert rel(a, b) < 1e-4
        assert g.generate(ids, GenerationConfig.greedy(16)) == s.generate(ids, GenerationConfig.greedy(16))
    finally:
        g.close(); s.close()


def test_group_sequences_and_batched_decode():
    """cm_seq_* / cm_decode_batch on the group handle: every rank keeps the same page tables; one all-reduce per projection
    covers all rows of the batch."""
    cfg = configs.get_config("tiny-qwen3-untied")
    g, s = _group(cfg, 2, max_seqs=8), _single(cfg, max_seqs=8)
    try:
        prompts = [configs.synthetic_prompt(n, cfg["vocab_size"]) for n in (9, 17, 5)]
        out = []
        for m in (g, s):
            sq = [m.seq_alloc() for _ in prompts]
            last = [m.seq_forward(q, p, 0, want_logits=False)[1] for q, p in zip(sq, prompts)]
            toks = [list(last)]
            for _ in range(6):
                _, nxt = m.step_batch_decode(sq, toks[-1], want_logits=False)
                toks.append([int(t) for t in nxt])
            f = m.seq_fork(sq[0])
            assert m.seq_len(f) == m.seq_len(sq[0])
            for q in sq + [f]:
                m.seq_free(q)
            out.append(toks)
        assert out[0] == out[1]
    finally:
        g.close(); s.close()


def test_group_engine_emits_the_tokens_of_the_unsharded_engine():
    """A cm_engine on the group handle: one scheduler per rank in lockstep, rank 0's events reported."""
    from crane_amd.engine import GenerationParams, InferenceEngine
    cfg = configs.get_config("tiny-qwen3-untied")
    outs = []
    for mk in (lambda: _group(cfg, 2, max_seqs=6), lambda: _single(cfg, max_seqs=6)):
        m = mk()
        try:
            e = InferenceEngine(m, max_running=4)
            ids = []
            for n, t in ((7, 0.0), (19, 0.0), (11, 0.7), (5, 0.0), (13, 0.9)):
                gp = GenerationParams.greedy(10) if t == 0.0 else GenerationParams(max_tokens=10, temperature=t, seed=3)
                ids.append(e.submit(configs.synthetic_prompt(n, cfg["vocab_size"]), gp))
            toks, done = e.run_until_idle()
            assert all(done[i].kind == "finished" for i in ids)
            outs.append([toks[i] for i in ids])
            e.close()
        finally:
            m.close()
    assert outs[0] == outs[1]


@pytest.mark.parametrize("tp", [4, 8])
def test_headline_geometry_two_layers(tp):
    """Qwen3-8B widths (H 4096, 32 / 8 heads, I 12288, V 151 936), 2 layers, TP = 4 and TP = 8 (4 q heads + 1 kv head +
    1536 MLP columns + 18 992 vocabulary rows per rank) on one device, f16 pages: the shard shapes of the real model through the
    group handle, against the TP = 1 handle."""
    cfg = dict(configs.get_config("qwen3-8b"), num_hidden_layers=2)
    ids = configs.synthetic_prompt(40, cfg["vocab_size"])
    g, s = _group(cfg, tp, max_seq_len=128, max_seqs=1), _single(cfg, max_seq_len=128, max_seqs=1, engine=-1)
    try:
        a, b = g.forward_step(ids, 0)[0, 0], s.forward_step(ids, 0)[0, 0]
        assert rel(a, b) < 1e-4
        tg = [g.forward_step_greedy([7], len(ids))]
        ts = [s.forward_step_greedy([7], len(ids))]
        for i in range(8):
            tg.append(g.forward_step_greedy([tg[-1]], len(ids) + 1 + i)); ts.append(s.forward_step_greedy([ts[-1]], len(ids) + 1 + i))
        assert tg == ts
    finally:
        g.close(); s.close()


@pytest.mark.parametrize("n", [2, 4, 8])
def test_peer_store_collectives_alone(n):
    """cm_debug_peer_selftest: n rank threads on one device, the all-reduce and the all-gather of csrc/kernels_tp.hip on vectors
    of a few sizes (one workgroup, several workgroups, a grid-stride tail), every sum (rank order) and every gathered word
    checked on the host, 20 rounds each -- epochs, parity double-buffering and the finish ticket of consecutive collectives."""
    from crane_amd import _lib
    lib = _lib.load()
    for count in (7, 512, 4096, 21 * 512):
        bad = lib.cm_debug_peer_selftest(n, 0, 20, count)
        assert bad == 0, (n, count, bad, lib.cm_last_global_error())


@pytest.mark.parametrize("n", [2, 8])
def test_peer_store_epochs_across_the_32_bit_wrap(n):
    """iters < 0 starts the ranks' epoch counter at 0xFFFFFFFD: the 12 rounds (24 collectives) walk 0xFFFFFFFE, 0xFFFFFFFF, 2, 3, ...
    -- the inbox buffer is the epoch's parity, so the wrap must keep consecutive epochs alternating (0 = never written and 1 are
    skipped; with 0xFFFFFFFF -> 1 two consecutive collectives shared a buffer and a fast rank overwrote granules a slow one had
    not read: the reader then spun into its 2 s bound)."""
    from crane_amd import _lib
    lib = _lib.load()
    for count in (512, 21 * 512):
        bad = lib.cm_debug_peer_selftest(n, 0, -12, count)
        assert bad == 0, (n, count, bad, lib.cm_last_global_error())


def test_groups_of_different_sizes_one_after_another():
    """Handles of different tensor-parallel degree created and destroyed in one process (recycled device memory, recycled
    worker threads): found a stale-data hazard of re-used uncached allocations during development (csrc/tp.cpp init_peer)."""
    from crane_amd.backend import GenerationConfig
    cfg = configs.get_config("tiny-qwen3-untied")
    ids = configs.synthetic_prompt(21, cfg["vocab_size"])
    s = _single(cfg)
    ref, rtok = s.forward_step(ids, 0)[0, 0], s.generate(ids, GenerationConfig.greedy(12))
    s.close()
    for tp in (4, 2, 2, 4, 2):
        g = _group(cfg, tp)
        try:
            assert rel(g.forward_step(ids, 0)[0, 0], ref) < 1e-4
            assert g.generate(ids, GenerationConfig.greedy(12)) == rtok
        finally:
            g.close()


@pytest.mark.parametrize("name", ["tiny-qwen3-untied", "tiny-qwen3.5"])
@pytest.mark.parametrize("fmt", ["q8_0", "q4_0"])
def test_isq_under_tensor_parallelism(name, fmt):
    """In-situ quantisation of a sharded model: every rank quantises ITS shard; the 32-weight blocks run along K and the row-parallel
    cuts (o_proj / out_proj on whole heads, down_proj on the intermediate slice) fall on block boundaries, so the codes are the
    unsharded model's codes and only the f32 sums across ranks differ in order.  Dense and hybrid (Gated-Delta-Net in_proj /
    out_proj quantised, a / b gate rows bf16), prompt + decode + greedy ids against the TP = 1 handle -- with f32 activations (exact
    dequantised weights: tight), and with the default integer-dot activations (an 8-bit activation code on a rounding boundary flips
    with the 1e-6 the summation order moves its input: the documented ~1e-2 of ANY two implementations)."""
    from crane_amd.backend import GenerationConfig
    cfg = configs.get_config(name)
    ids = configs.synthetic_prompt(21, cfg["vocab_size"])
    for act, tol in (("f32", 1e-4), ("int", 3e-2)):
        g, s = _group(cfg, 2, isq=fmt, quant_act=act), _single(cfg, isq=fmt, quant_act=act)
        try:
            a, b = g.forward_step(ids, 0)[0, 0], s.forward_step(ids, 0)[0, 0]
            assert rel(a, b) < tol, (act, rel(a, b))
            a, b = g.forward_step([5], len(ids))[0, 0], s.forward_step([5], len(ids))[0, 0]
            assert rel(a, b) < tol, (act, rel(a, b))
            g.clear_kv_cache(); s.clear_kv_cache()
            g.debug_set("no_prefill", 1); s.debug_set("no_prefill", 1)          # the prompt through the quantised decode kernels
            a, b = g.forward_step(ids[:9], 0)[0, 0], s.forward_step(ids[:9], 0)[0, 0]
            assert rel(a, b) < tol, (act, rel(a, b))
            if act == "f32":
                assert g.generate(ids, GenerationConfig.greedy(12)) == s.generate(ids, GenerationConfig.greedy(12))
        finally:
            g.close(); s.close()


def test_gguf_checkpoint_under_tensor_parallelism(tmp_path):
    """A llama.cpp-style mixed-type GGUF file (Q4_K / Q6_K / Q8_0 matrices, quantised embedding) loaded by a TP = 2 group: the
    loader cuts the quantised tensors by row (heads, gate / up, vocabulary) and by column on whole ggml blocks; f32 activations so
    that only the order of the cross-rank sums differs from the TP = 1 handle."""
    from crane_amd.backend import GenerationConfig, Model
    from oracle import gguf_oracle as G
    cfg = configs.get_config("tiny-qwen3-untied")
    w = synth.synth_weights_f32(cfg, seed=0)

    def mixed(name, shape):
        if "attn_v" in name or "ffn_up" in name or name == "token_embd.weight":
            return G.GGML_Q6_K
        if "attn_output" in name or name == "output.weight":
            return G.GGML_Q8_0
        return G.GGML_Q4_K
    path = str(tmp_path / "tp.gguf")
    G.write_qwen3_gguf(path, cfg, w, mixed)
    ids = configs.synthetic_prompt(21, cfg["vocab_size"])
    kw = dict(max_seq_len=128, max_seqs=2, quant_act="f32")
    g = Model.from_pretrained(path, tp_size=2, tp_in_process=True, tp_devices=[0, 0], **kw)
    s = Model.from_pretrained(path, **kw)
    try:
        a, b = g.forward_step(ids, 0)[0, 0], s.forward_step(ids, 0)[0, 0]
        assert rel(a, b) < 1e-4
        a, b = g.forward_step([5], len(ids))[0, 0], s.forward_step([5], len(ids))[0, 0]
        assert rel(a, b) < 1e-4
        assert g.generate(ids, GenerationConfig.greedy(10)) == s.generate(ids, GenerationConfig.greedy(10))
    finally:
        g.close(); s.close()


def test_int8_matrix_core_projections_under_tensor_parallelism():
    """ISQ Q8_0 at the 8B widths (2 layers) on a TP = 2 group (round 6): the prompt pass and the decode groups of 8 or more sequences run
    every rank's projections on the int8 matrix cores -- row-parallel o_proj / down_proj as partial sums over the rank's K slice (whole
    32-weight blocks), ONE all-reduce per projection for all rows, the norm + quantiser of the next projection after it.  Against the
    TP = 1 handle: same codes, sums across ranks in another order => the documented bound of two integer-dot implementations (an
    activation code on a rounding boundary flips with the last bit of its input, DESIGN 3.9: 3e-2 of the logit range), and most
    greedy ids equal."""
    cfg = configs.get_config("qwen3-8b-2l")
    V = cfg["vocab_size"]
    g, s = _group(cfg, 2, isq="q8_0", max_seqs=18), _single(cfg, isq="q8_0", max_seqs=18)
    try:
        ids = [(11 * i + 5) % V for i in range(200)]
        a, b = g.forward_step(ids, 0)[0, 0], s.forward_step(ids, 0)[0, 0]
        assert rel(a, b) < 3e-2, rel(a, b)
        outs = []
        for m in (g, s):
            sq = [m.seq_alloc() for _ in range(16)]
            for i, q in enumerate(sq):
                m.seq_forward(q, [(7 * k + 3 + 11 * i) % V for k in range(20 + i)], 0, want_logits=False)
            lg, nxt = m.step_batch_decode(sq, [(5 + 3 * i) % V for i in range(16)])       # (the same tokens on both handles)
            outs.append((lg[:, 0], nxt))
        for i in range(16):
            assert rel(outs[0][0][i], outs[1][0][i]) < 3e-2, (i, rel(outs[0][0][i], outs[1][0][i]))
        assert int((outs[0][1] == outs[1][1]).sum()) >= 12
    finally:
        g.close(); s.close()


@pytest.mark.parametrize("name", ["tiny-qwen3-vl", "tiny-qwen3.5-vl"])
def test_vision_language_group(name):
    """Image + text through a TP = 2 group: the tower is replicated (every rank encodes), the image rows are spliced into every
    rank's hidden rows, DeepStack maps are added on every rank, 3-axis MRoPE positions and rope_delta are per-rank copies."""
    from crane_amd.processor import PreprocessorConfig
    cfg = configs.get_config(name)
    rng = np.random.default_rng(0)
    image = rng.integers(0, 256, size=(64, 96, 3), dtype=np.uint8)
    vc = cfg["vision_config"]
    pix, grid = PreprocessorConfig(patch_size=vc.get("patch_size", 16), temporal_patch_size=vc.get("temporal_patch_size", 2),
                                   merge_size=vc.get("spatial_merge_size", 2)).process(image)
    g, s = _group(cfg, 2), _single(cfg)
    try:
        fa, fb = g.encode_images(pix, [list(grid)]), s.encode_images(pix, [list(grid)])
        assert np.array_equal(fa, fb)                                   # the tower is not sharded
        img = cfg["image_token_id"]
        ids = [5, 6, cfg["vision_start_token_id"]] + [img] * int(fa.shape[0]) + [cfg["vision_end_token_id"], 8, 9]
        (la, ta), (lb, tb) = g.vlm_forward(ids, pix, [list(grid)]), s.vlm_forward(ids, pix, [list(grid)])
        assert rel(la, lb) < 1e-4 and ta == tb
        toks_g, toks_s, pos = [ta], [tb], len(ids)
        for i in range(6):
            toks_g.append(g.forward_step_greedy([toks_g[-1]], pos + i)); toks_s.append(s.forward_step_greedy([toks_s[-1]], pos + i))
        assert toks_g == toks_s
    finally:
        g.close(); s.close()


def test_large_decode_group_under_tensor_parallelism():
    """20 sequences per decode round: the projections run as MFMA GEMMs over the rows of the batch, ONE all-reduce per projection
    covers all rows, the vocabulary-sharded head feeds one gather; int8 KV pages on top (codes and scales sharded with their heads)."""
    cfg = configs.get_config("tiny-qwen3-untied")
    for kv in ("f16", "int8"):
        g, s = _group(cfg, 2, max_seqs=24, kv_dtype=kv), _single(cfg, max_seqs=24, kv_dtype=kv)
        try:
            outs = []
            for m in (g, s):
                sq = [m.seq_alloc() for _ in range(20)]
                last = [m.seq_forward(q, configs.synthetic_prompt(5 + (i % 7), cfg["vocab_size"]), 0, want_logits=False)[1] for i, q in enumerate(sq)]
                toks = [list(last)]
                for _ in range(4):
                    _, nxt = m.step_batch_decode(sq, toks[-1], want_logits=False)
                    toks.append([int(t) for t in nxt])
                outs.append(toks)
            assert outs[0] == outs[1], kv
        finally:
            g.close(); s.close()


@pytest.mark.parametrize("kind", ["q8_0", "mixed35"])
def test_hybrid_gguf_checkpoint_under_tensor_parallelism(tmp_path, kind):
    """llama.cpp `qwen35` GGUF (CHUNKED value-head order: a rank's value heads are strided in the file) loaded by a TP = 2 group:
    q / k / v conv channels, z, the bf16 b / a rows, A_log, dt_bias gathered by head, ssm_out cut by column in runs of NK_l value
    heads, the gated attention's per-head [q | gate] rows, mixed ggml types inside in_proj / q|k|v / gate|up."""
    from crane_amd.backend import GenerationConfig, Model
    from oracle import gguf_oracle as G
    cfg = configs.get_config("tiny-qwen3.5")
    w = synth.synth_weights_f32(cfg, seed=0)
    if kind == "q8_0":
        type_of = lambda name, shape: G.GGML_Q8_0
    else:
        def type_of(name, shape):
            if "attn_gate" in name or "attn_v" in name or "ffn_up" in name or name == "token_embd.weight":
                return G.GGML_Q6_K
            if "ssm_out" in name or "attn_output" in name or name == "output.weight" or "ffn_down" in name:
                return G.GGML_Q8_0            # column cuts of 128 / 256 columns: whole 32-weight blocks
            return G.GGML_Q4_K
    path = str(tmp_path / f"tp35-{kind}.gguf")
    G.write_qwen35_gguf(path, cfg, w, type_of)
    ids = configs.synthetic_prompt(21, cfg["vocab_size"])
    kw = dict(max_seq_len=128, max_seqs=2, quant_act="f32")
    g = Model.from_pretrained(path, tp_size=2, tp_in_process=True, tp_devices=[0, 0], **kw)
    s = Model.from_pretrained(path, **kw)
    try:
        a, b = g.forward_step(ids, 0)[0, 0], s.forward_step(ids, 0)[0, 0]
        assert rel(a, b) < 1e-4
        a, b = g.forward_step([5], len(ids))[0, 0], s.forward_step([5], len(ids))[0, 0]
        assert rel(a, b) < 1e-4
        assert g.generate(ids, GenerationConfig.greedy(10)) == s.generate(ids, GenerationConfig.greedy(10))
    finally:
        g.close(); s.close()
