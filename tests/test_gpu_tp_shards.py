"""One TP rank at a time on ONE GPU (cm_opts.debug_flags = CM_DEBUG_TP_LOCAL: collectives are local no-ops): the C++ loader's shard of every
weight + the kernels running on local head counts must reproduce the oracle evaluated on the SAME shard with an
identity all-reduce.  Together with the gloo tests (real all-reduce on the same plan) this covers the TP path that
cannot be run on a single-GPU box."""
import os

import numpy as np
import pytest

from crane_amd import configs, synth, tp

pytestmark = pytest.mark.gpu


def rel(a, ref):
    return float(np.abs(a - ref).max() / np.abs(ref).max())


def _run_rank(cfg, rank, world, ids, isq=None, **live):
    from crane_amd.backend import Model
    m = Model.synthetic(cfg, seed=0, max_seq_len=128, max_seqs=2, kv_dtype="f32", tp_rank=rank, tp_size=world,
                        tp_unique_id=b"\0" * 128, isq=isq, debug_tp_local=True, **live)
    try:
        a = m.forward_step(ids, 0)[0, 0]
        b = m.forward_step([5], len(ids))[0, 0]
        return a, b
    finally:
        m.close()


@pytest.mark.parametrize("world", [2, 4])
def test_dense_rank_shards(world):
    from oracle.qwen3_oracle import Qwen3Config, Qwen3Oracle
    cfg = configs.get_config("tiny-qwen3-untied")                     # 8 q heads, 2 kv heads (replicated at tp=4)
    w = synth.synth_weights_f32(cfg, 0)
    ids = configs.synthetic_prompt(21, cfg["vocab_size"])
    for rank in range(world):
        plan = tp.shard_plan(cfg, world, rank)
        sw = tp.shard_weights(cfg, w, plan)
        local = dict(cfg, num_attention_heads=len(plan.q_heads), num_key_value_heads=len(plan.kv_heads),
                     intermediate_size=len(plan.inter))
        o = Qwen3Oracle(Qwen3Config.from_json(local), sw)
        got_a, got_b = _run_rank(cfg, rank, world, ids)
        v = slice(plan.vocab.start, plan.vocab.stop)
        assert rel(got_a[v], o.forward(ids, 0)) < 1e-4 and rel(got_b[v], o.forward([5], len(ids))) < 1e-4


@pytest.mark.parametrize("world,ranks", [(8, (0, 7)), (4, (1,)), (2, (1,))])
def test_dense_rank_shard_on_the_persistent_kernel(world, ranks):
    """cm_opts.engine = 1 on ONE rank's shard (CM_DEBUG_TP_LOCAL: the exchange is the identity, so the rank's step is a TP = 1 step at
    the shard's widths): the whole-token persistent kernel at 512-element dependency chunks (TP = 8 of the 2048 / 4096-wide test
    model: Hq_l * D = 512, I_l = 512 -- staging passes with a padded upper half), 1024 (TP = 4) and 2048 (TP = 2), the attention of
    ONE kv head on 32 of the workgroups (TP = 8), embedding row + vocabulary-sharded head + arg-max partials with a global row
    index inside the launch.  Decode steps over a cache the prompt pass wrote (f16 pages), against the f32 oracle on the same
    shard (bar 1e-3) and against the launch path of the same handle; the device-chained greedy loop emits the launch path's ids."""
    from crane_amd.backend import Model
    from oracle.qwen3_oracle import Qwen3Config, Qwen3Oracle
    cfg = configs.get_config("eng-qwen3")
    w = synth.synth_weights_f32(cfg, 0)
    ids = configs.synthetic_prompt(37, cfg["vocab_size"])
    for rank in ranks:
        plan = tp.shard_plan(cfg, world, rank)
        sw = tp.shard_weights(cfg, w, plan)
        local = dict(cfg, num_attention_heads=len(plan.q_heads), num_key_value_heads=len(plan.kv_heads), intermediate_size=len(plan.inter))
        o = Qwen3Oracle(Qwen3Config.from_json(local), sw)
        v = slice(plan.vocab.start, plan.vocab.stop)
        m = Model.synthetic(cfg, seed=0, max_seq_len=256, max_seqs=2, tp_rank=rank, tp_size=world, tp_unique_id=b"\0" * 128,
                            debug_tp_local=True, engine=1)
        try:
            assert m.engine_active() == 2, "whole-token persistent kernel not active on the shard"
            ref = o.forward(ids, 0)
            assert rel(m.forward_step(ids, 0)[0, 0][v], ref) < 1e-3
            tok, steps = 5, []
            for i in range(4):
                ref = o.forward([tok], len(ids) + i)
                got = m.forward_step([tok], len(ids) + i)[0, 0][v]
                assert rel(got, ref) < 1e-3, (world, rank, i, rel(got, ref))
                steps.append(got.copy())
                tok = (7 * tok + 3) % cfg["vocab_size"]
            m.debug_set("engine", 0)                       # the same steps on the per-projection launches
            m.clear_kv_cache(); m.forward_step(ids, 0)
            tok = 5
            for i in range(4):
                got = m.forward_step([tok], len(ids) + i)[0, 0][v]
                assert rel(steps[i], got) < 1e-4, (world, rank, i, rel(steps[i], got))
                tok = (7 * tok + 3) % cfg["vocab_size"]
            # the device-chained greedy loop (rank-local arg-max over the rank's vocabulary rows, global indices)
            outs = []
            for eng in (0, 1):
                m.debug_set("engine", eng)
                m.clear_kv_cache(); m.forward_step(ids, 0)
                toks, _ = m.bench_decode(5, 6)
                outs.append([int(t) for t in toks])
            assert outs[0] == outs[1] and all(plan.vocab.start <= t < plan.vocab.stop for t in outs[1]), outs
        finally:
            m.close()


def test_hybrid_rank_shards():
    from oracle import qwen3_5_oracle as O5
    cfg = configs.get_config("tiny-qwen3.5")
    w = synth.synth_weights_f32(cfg, 0)
    ids = configs.synthetic_prompt(21, cfg["vocab_size"])
    for rank in range(2):
        plan = tp.shard_plan(cfg, 2, rank)
        sw = tp.shard_weights(cfg, w, plan)
        local = dict(cfg, num_attention_heads=len(plan.q_heads), num_key_value_heads=len(plan.kv_heads),
                     linear_num_key_heads=len(plan.gdn_key_heads), linear_num_value_heads=len(plan.gdn_value_heads),
                     tie_word_embeddings=True)
        sw["model.embed_tokens.weight"] = w["model.embed_tokens.weight"]
        o = O5.Qwen35Oracle(O5.Qwen35Config.from_json(local), sw)
        o.lm_head = w["lm_head.weight"][plan.vocab.start:plan.vocab.stop]
        got_a, got_b = _run_rank(cfg, rank, 2, ids)
        v = slice(plan.vocab.start, plan.vocab.stop)
        assert rel(got_a[v], o.forward(ids, 0)) < 1e-4 and rel(got_b[v], o.forward([5], len(ids))) < 1e-4


def test_dense_rank_shards_isq_q8_0(monkeypatch):
    """ISQ under TP: every rank quantises its own shard (Q8_0 blocks never straddle a shard boundary), the row-parallel
    projections feed the all-reduce, the vocabulary-sharded quantised lm_head feeds the arg-max gather."""
    from oracle import gguf_oracle as G
    from oracle.qwen3_oracle import Qwen3Config, Qwen3Oracle
    cfg = configs.get_config("tiny-qwen3-untied")
    w = synth.synth_weights_f32(cfg, 0)
    ids = configs.synthetic_prompt(21, cfg["vocab_size"])
    world = 2
    linears = ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj")
    for rank in range(world):
        plan = tp.shard_plan(cfg, world, rank)
        sw = tp.shard_weights(cfg, w, plan)
        for k, v in list(sw.items()):
            if any(k.endswith(f"{l}.weight") for l in linears) or k == "lm_head.weight":
                sw[k] = G.dequantize_q8_0(G.quantize_q8_0(v), v.size).reshape(v.shape)
        local = dict(cfg, num_attention_heads=len(plan.q_heads), num_key_value_heads=len(plan.kv_heads),
                     intermediate_size=len(plan.inter))
        o = Qwen3Oracle(Qwen3Config.from_json(local), sw)
        got_a, got_b = _run_rank(cfg, rank, world, ids, isq="q8_0", quant_act="f32", quant_prefill=False)
        v = slice(plan.vocab.start, plan.vocab.stop)
        assert rel(got_a[v], o.forward(ids, 0)) < 2e-4 and rel(got_b[v], o.forward([5], len(ids))) < 2e-4


@pytest.mark.parametrize("name,nb,isq", [("tiny-qwen3-untied", 2, None), ("tiny-qwen3-untied", 5, None), ("tiny-qwen3.5", 3, None),
                                         ("tiny-qwen3-untied", 5, "q8_0"), ("tiny-qwen3-untied", 21, None), ("tiny-qwen3.5", 19, None)])
def test_batched_decode_on_a_rank(name, nb, isq):
    """cm_decode_batch under TP: the row-parallel projections of all sequences go through ONE all-reduce per layer and the
    vocabulary-sharded lm_head through one gather.  With CM_DEBUG_TP_LOCAL a rank's batched step must agree with its own
    single-sequence steps on its vocabulary slice (logits and rank-local arg-max)."""
    from crane_amd.backend import Model
    cfg = configs.get_config(name)
    V = cfg["vocab_size"]
    world = 2
    for rank in range(world):
        plan = tp.shard_plan(cfg, world, rank)
        v = slice(plan.vocab.start, plan.vocab.stop)
        m = Model.synthetic(cfg, seed=0, max_seq_len=128, max_seqs=nb + 2, kv_dtype="f32", tp_rank=rank, tp_size=world,
                            tp_unique_id=b"\0" * 128, isq=isq, debug_tp_local=True)
        try:
            seqs, toks = [], []
            for b in range(nb):
                s = 0 if b == 0 else m.seq_alloc()
                _, g = m.seq_forward(s, [(7 * i + 3 + 11 * b) % V for i in range(5 + 3 * (b % 12))], 0, want_logits=False)
                seqs.append(s); toks.append(int(g))
            for r in range(2):
                want = []
                for s, t in zip(seqs, toks):
                    f = m.seq_fork(s)
                    lg, g = m.seq_forward(f, [t], m.seq_len(f))
                    want.append((lg.reshape(-1)[v].copy(), int(g)))
                    m.seq_free(f)
                lg, greedy = m.step_batch_decode(seqs, toks)
                for b in range(nb):
                    # (17+ sequences: the projections are MFMA GEMMs over the batch rows -- another summation order)
                    assert rel(lg[b, 0][v], want[b][0]) < (3e-5 if nb < 17 else 1e-4), (rank, r, b)
                    assert int(greedy[b]) == want[b][1]
                toks = [int(g) for g in greedy]
        finally:
            m.close()


@pytest.mark.parametrize("kind", ["q8_0", "q4_k", "mixed"])
def test_dense_rank_shards_gguf(tmp_path, monkeypatch, kind):
    """GGUF checkpoints under TP: the loader cuts the quantised tensors by row (q / kv heads, gate / up, vocabulary) and by
    COLUMN on whole ggml blocks (o_proj, down_proj: 32 for Q8_0, 256 for the K-quants), so a rank's codes are the file's
    codes.  Checked with f32 activations against the f32 oracle on the dequantised shard."""
    from crane_amd.backend import Model
    from oracle import gguf_oracle as G
    from oracle.qwen3_oracle import Qwen3Config, Qwen3Oracle
    from tests.test_gpu_quant import _types
    cfg = configs.get_config("tiny-qwen3-untied")            # per rank at tp = 2: 512 o_proj columns, 768 down_proj columns
    w = synth.synth_weights_f32(cfg, 0)
    path = str(tmp_path / f"tp-{kind}.gguf")
    deq, _ = G.write_qwen3_gguf(path, cfg, w, _types(kind), want_qmats=True)
    ids = configs.synthetic_prompt(21, cfg["vocab_size"])
    world = 2
    for rank in range(world):
        plan = tp.shard_plan(cfg, world, rank)
        sw = tp.shard_weights(cfg, deq, plan)
        local = dict(cfg, num_attention_heads=len(plan.q_heads), num_key_value_heads=len(plan.kv_heads),
                     intermediate_size=len(plan.inter))
        o = Qwen3Oracle(Qwen3Config.from_json(local), sw)
        m = Model.from_pretrained(path, max_seq_len=128, max_seqs=2, kv_dtype="f32", tp_rank=rank, tp_size=world,
                                  tp_unique_id=b"\0" * 128, debug_tp_local=True, quant_act="f32", quant_prefill=False)
        try:
            got_a = m.forward_step(ids, 0)[0, 0]
            got_b = m.forward_step([5], len(ids))[0, 0]
        finally:
            m.close()
        v = slice(plan.vocab.start, plan.vocab.stop)
        assert rel(got_a[v], o.forward(ids, 0)) < 2e-4 and rel(got_b[v], o.forward([5], len(ids))) < 2e-4
