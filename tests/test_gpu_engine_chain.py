"""The persistent decode kernel (kernels_engine.hip, cm_opts.engine = 1) in both of its modes -- "full": every projection
and the attention of every layer of a token in ONE launch; "layer": one launch per layer for o_proj -> gate||up ->
down_proj -> next QKV around the separate attention kernels -- against (a) the per-projection launch path it replaces
(same arithmetic up to summation order) and (b) the numpy oracle of the reference forward.
"""
import os
import numpy as np
import pytest

from crane_amd import configs, synth
from crane_amd.backend import GenerationConfig, Model
from oracle.qwen3_oracle import Qwen3Config, Qwen3Oracle

pytestmark = pytest.mark.gpu


def rel(a, ref):
    return float(np.abs(a - ref).max() / np.abs(ref).max())


@pytest.fixture(scope="module", params=[("full", "f16", "eng-qwen3"), ("layer", "f16", "eng-qwen3"), ("full", "bf16", "eng-qwen3"),
                                        ("full", "f16", "eng-qwen3-gqa2"), ("full", "f16", "eng-qwen3-h1024"),
                                        ("layer", "f16", "eng-qwen3-h1024")], ids=lambda p: f"{p[0]}-{p[1]}-{p[2]}")
def trio(request):
    mode, kv, name = request.param
    cfg = configs.get_config(name)
    w = synth.synth_weights_f32(cfg, seed=0)
    eng = Model.synthetic(cfg, seed=0, max_seq_len=2048, max_seqs=2, engine=1, kv_dtype=kv)
    eng.debug_set("engine_full", 1 if mode == "full" else 0)          # whole-token launch / one launch per layer
    assert eng.engine_active() == (2 if mode == "full" else 1)
    eng.kv = kv
    ref = Model.synthetic(cfg, seed=0, max_seq_len=2048, max_seqs=2, engine=-1, kv_dtype=kv)
    yield cfg, w, eng, ref
    eng.close()
    ref.close()


def test_chain_equals_launch_path_and_oracle(trio):
    cfg, w, eng, ref = trio
    o = Qwen3Oracle(Qwen3Config.from_json(cfg), w, kv_dtype=eng.kv)
    o32 = Qwen3Oracle(Qwen3Config.from_json(cfg), w)                  # the pure f32 CPU forward
    ids = configs.synthetic_prompt(12, cfg["vocab_size"])
    for m in (eng, ref):
        m.clear_kv_cache()
    for pos, t in enumerate(ids):                         # token-serial: every step is a decode step
        a = eng.forward_step([t], pos)[0, 0]
        b = ref.forward_step([t], pos)[0, 0]
        c = o.forward([t], pos)
        assert rel(a, b) < 1e-4, (pos, rel(a, b))      # the sum of squares of the RMSNorm is accumulated in another order;
                                                         # a K/V element on a bf16 tie then rounds the other way (2^-9 on it)
        assert rel(a, c) < 1e-3, (pos, rel(a, c))      # bf16 K/V appends: a value on a rounding tie moves by 2^-9
        assert int(a.argmax()) == int(c.argmax())
        d = o32.forward([t], pos)
        if eng.kv == "f16":                              # the default pages: north_star's bar against the unrounded forward
            assert rel(a, d) < 1e-3, (pos, rel(a, d))


def test_chain_generate_and_graph_replay(trio):
    cfg, w, eng, ref = trio
    ids = configs.synthetic_prompt(9, cfg["vocab_size"])
    want = ref.generate(ids, GenerationConfig.greedy(24))
    assert eng.generate(ids, GenerationConfig.greedy(24)) == want
    assert eng.generate(ids, GenerationConfig.greedy(24), sync_every=8) == want      # hipGraph replays back to back


def test_chain_long_context_and_bench_loop(trio):
    """context >= 768: the per-layer mode switches to the MFMA flash-decode kernel; the full mode's own split-KV attention
    walks more than two chunks per workgroup"""
    cfg, w, eng, ref = trio
    outs = []
    for m in (eng, ref):
        m.debug_fill_kv(1000, seed=3)
        toks, _ = m.bench_decode(5, 40)
        lg = m.forward_step([7], 1040)[0, 0]
        outs.append(([int(t) for t in toks], lg))
    assert outs[0][0] == outs[1][0]
    assert rel(outs[0][1], outs[1][1]) < 1e-4


@pytest.mark.parametrize("name", ["eng-qwen3-h1024", "eng-qwen3-gqa2", "eng-qwen3"])
@pytest.mark.parametrize("mode", [0, 1])
def test_graph_replay_as_the_first_use_of_a_handle(name, mode):
    """The device-chained greedy loop (hipGraph capture + replays) as the FIRST thing a handle does, in the per-layer and the
    whole-token mode: every kernel instantiation a handle can launch is prepared at cm_create (an instantiation whose first
    launch happened inside a stream capture timed out on its first replay)."""
    cfg = configs.get_config(name)
    outs = []
    for engine in (1, -1):
        m = Model.synthetic(cfg, seed=0, max_seq_len=2048, max_seqs=1, engine=engine)
        try:
            if engine == 1:
                m.debug_set("engine_full", mode)
                assert m.engine_active() == (2 if mode else 1)
            m.debug_fill_kv(300, seed=2)
            toks, _ = m.bench_decode(3, 12)
            outs.append([int(t) for t in toks])
            if engine == 1:
                assert m.engine_active() == (2 if mode else 1)        # (no timeout + fallback happened)
        finally:
            m.close()
    assert outs[0] == outs[1]


@pytest.mark.parametrize("ctx", [2500, 5000])
def test_chain_context_beyond_the_prefetched_chunks(ctx):
    """whole-token mode: a workgroup prefetches 4 K/V chunks per layer (contexts up to 2048); longer contexts walk the rest
    in the generic two-chunks-per-iteration loop -- same tokens and logits as the launch path"""
    cfg = configs.get_config("eng-qwen3")
    outs = []
    for engine in (1, -1):
        m = Model.synthetic(cfg, seed=0, max_seq_len=ctx + 128, max_seqs=1, engine=engine)      # (f16 pages, the default)
        try:
            m.debug_fill_kv(ctx, seed=5)
            toks, _ = m.bench_decode(5, 12)
            lg = m.forward_step([7], ctx + 12)[0, 0]
            outs.append(([int(t) for t in toks], lg))
        finally:
            m.close()
    assert outs[0][0] == outs[1][0]
    assert rel(outs[0][1], outs[1][1]) < 1e-4


@pytest.mark.parametrize("name", ["eng-qwen3", "eng-qwen3-gqa2"])
@pytest.mark.parametrize("ctx", [500, 2036])
def test_chain_steps_across_the_split_round_boundaries(name, ctx):
    """whole-token mode, round 6 (attention in two parts: the old tokens from the pages, the step's own token folded in by the merging
    wave): steps that carry the context across a split round (32 splits x 16 tokens = 512: the new token opens chunk 1 of split 0) and
    across the prefetched chunks (4 x 512 = 2048) -- tokens and logits of the launch path at every position"""
    cfg = configs.get_config(name)
    outs = []
    for engine in (1, -1):
        m = Model.synthetic(cfg, seed=0, max_seq_len=ctx + 128, max_seqs=1, engine=engine)
        try:
            if engine == 1:
                assert m.engine_active() == 2
            m.debug_fill_kv(ctx, seed=9)
            tok, lgs = 11, []
            for pos in range(ctx, ctx + 30):                     # 500 .. 529 / 2036 .. 2065
                lg = m.forward_step([tok], pos)[0, 0]
                lgs.append(lg.copy())
                tok = int(lg.argmax())
            outs.append(lgs)
        finally:
            m.close()
    for i, (a, b) in enumerate(zip(*outs)):
        assert rel(a, b) < 1e-4, (ctx + i, rel(a, b))
        assert int(a.argmax()) == int(b.argmax())


def test_engine_refused_when_shapes_do_not_fit():
    from crane_amd._lib import CraneError
    cfg = configs.get_config("tiny-qwen3")               # hidden 256: not a multiple of 2048
    with pytest.raises(CraneError):
        Model.synthetic(cfg, seed=0, max_seq_len=128, max_seqs=2, engine=1)
    m = Model.synthetic(cfg, seed=0, max_seq_len=128, max_seqs=2, engine=0)      # default: falls back to the launches
    m.close()
