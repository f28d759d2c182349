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
