"""Tensor-parallel plan: world_size-2 (and 4) gloo runs on CPU.

Each rank slices the synthetic checkpoint with crane_amd.tp.shard_plan / shard_weights (the same
arithmetic csrc/loader.cpp uses), runs the oracle's layer math on its shard, all-reduces the
row-parallel partial sums with gloo, all-gathers the vocab-sharded logits -- and must reproduce the
unsharded oracle.  Correct-by-construction evidence for the RCCL path, which cannot be run here.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, cfg_name, qh):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from crane_amd import configs, synth, tp
    from oracle import qwen3_oracle as O
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = configs.get_config(cfg_name)
        w = synth.synth_weights_f32(cfg, 0)
        plan = tp.shard_plan(cfg, world, rank)
        sw = tp.shard_weights(cfg, w, plan)
        c = O.Qwen3Config.from_json(cfg)
        D, H = c.hd, c.hidden_size
        ids = configs.synthetic_prompt(9, c.vocab_size)
        cos, sin = O.rotary_tables(D, 64, c.rope_theta)
        S = len(ids)
        x = w["model.embed_tokens.weight"][np.asarray(ids)].astype(np.float32)

        def allreduce(a):
            t = torch.from_numpy(np.ascontiguousarray(a))
            dist.all_reduce(t)
            return t.numpy()

        for li in range(c.num_hidden_layers):
            p = f"model.layers.{li}."
            xn = O.rms_norm(x, w[p + "input_layernorm.weight"], c.rms_norm_eps)
            hq, hkv = len(plan.q_heads), len(plan.kv_heads)
            q = (xn @ sw[p + "self_attn.q_proj.weight"].T).reshape(S, hq, D)
            k = (xn @ sw[p + "self_attn.k_proj.weight"].T).reshape(S, hkv, D)
            v = (xn @ sw[p + "self_attn.v_proj.weight"].T).reshape(S, hkv, D)
            q = O.rope_thd(O.rms_norm(q, w[p + "self_attn.q_norm.weight"], c.rms_norm_eps), cos[:S], sin[:S])
            k = O.rope_thd(O.rms_norm(k, w[p + "self_attn.k_norm.weight"], c.rms_norm_eps), cos[:S], sin[:S])
            att = O.naive_gqa_attention(q.transpose(1, 0, 2), k.transpose(1, 0, 2), v.transpose(1, 0, 2),
                                        1.0 / np.sqrt(D), causal_offset=0)          # local heads only
            att = att.transpose(1, 0, 2).reshape(S, hq * D)
            x = x + allreduce(att @ sw[p + "self_attn.o_proj.weight"].T)            # all-reduce #1
            xn = O.rms_norm(x, w[p + "post_attention_layernorm.weight"], c.rms_norm_eps)
            h = O.silu(xn @ sw[p + "mlp.gate_proj.weight"].T) * (xn @ sw[p + "mlp.up_proj.weight"].T)
            x = x + allreduce(h @ sw[p + "mlp.down_proj.weight"].T)                 # all-reduce #2
        last = O.rms_norm(x[-1:], w["model.norm.weight"], c.rms_norm_eps)
        head = sw.get("lm_head.weight")
        if head is None:                                                            # tied: rows of the embedding
            head = w["model.embed_tokens.weight"][plan.vocab.start:plan.vocab.stop]
        local = (last @ head.T)[0]
        v_l = (c.vocab_size + world - 1) // world
        buf = np.zeros(v_l, np.float32); buf[:local.size] = local
        gathered = [torch.zeros(v_l) for _ in range(world)]
        dist.all_gather(gathered, torch.from_numpy(buf))
        logits = np.concatenate([g.numpy() for g in gathered])[:c.vocab_size]
        if rank == 0:
            ref = O.Qwen3Oracle(c, w).forward(ids, 0)
            qh.put(float(np.abs(logits - ref).max() / np.abs(ref).max()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("cfg_name,world", [("tiny-qwen3", 2), ("tiny-qwen3-untied", 2), ("tiny-qwen3-untied", 4)])
def test_tp_plan_reproduces_unsharded_forward(cfg_name, world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    qh = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, cfg_name, qh)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    err = qh.get(timeout=5)
    assert err < 1e-5, err


def test_plan_shapes_for_headline_models():
    from crane_amd import configs, tp
    c8 = configs.get_config("qwen3-8b")
    p = tp.shard_plan(c8, 8, 3)
    assert list(p.q_heads) == [12, 13, 14, 15] and list(p.kv_heads) == [3] and p.n_rep == 4
    assert len(p.inter) == 1536 and len(p.vocab) == 18992
    # Hkv=4 < tp=8 -> KV head r//2 replicated on two ranks, 3 q heads each (27B geometry)
    c27 = dict(c8, num_attention_heads=24, num_key_value_heads=4, intermediate_size=17408, vocab_size=248320)
    for r in range(8):
        p = tp.shard_plan(c27, 8, r)
        assert list(p.kv_heads) == [r // 2] and list(p.q_heads) == [3 * r, 3 * r + 1, 3 * r + 2] and p.n_rep == 3
        assert all(h // 6 == r // 2 for h in p.q_heads)          # every local q head maps to the local KV head
    with pytest.raises(ValueError):
        tp.shard_plan(c8, 3, 0)


def _worker35(rank, world, port, qh):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from crane_amd import configs, synth, tp
    from oracle import qwen3_5_oracle as O5
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = configs.get_config("tiny-qwen3.5")
        w = synth.synth_weights_f32(cfg, 0)
        plan = tp.shard_plan(cfg, world, rank)
        sw = tp.shard_weights(cfg, w, plan)
        local = dict(cfg, num_attention_heads=len(plan.q_heads), num_key_value_heads=len(plan.kv_heads),
                     linear_num_key_heads=len(plan.gdn_key_heads), linear_num_value_heads=len(plan.gdn_value_heads),
                     tie_word_embeddings=True)               # local head = this rank's vocab rows, gathered below
        sw["model.embed_tokens.weight"] = w["model.embed_tokens.weight"]
        o = O5.Qwen35Oracle(O5.Qwen35Config.from_json(local), sw)
        o.lm_head = w["lm_head.weight"][plan.vocab.start:plan.vocab.stop]

        def allreduce(a):
            t = torch.from_numpy(np.ascontiguousarray(a))
            dist.all_reduce(t)
            return t.numpy()
        o.allreduce = allreduce
        ids = configs.synthetic_prompt(9, cfg["vocab_size"])
        outs = [o.forward(ids, 0), o.forward([5], 9)]          # prefill chunk + one decode step (state hand-over)
        v_l = (cfg["vocab_size"] + world - 1) // world
        full = []
        for local_logits in outs:
            buf = np.zeros(v_l, np.float32); buf[:local_logits.size] = local_logits
            gathered = [torch.zeros(v_l) for _ in range(world)]
            dist.all_gather(gathered, torch.from_numpy(buf))
            full.append(np.concatenate([g.numpy() for g in gathered])[:cfg["vocab_size"]])
        if rank == 0:
            ref = O5.Qwen35Oracle(O5.Qwen35Config.from_json(cfg), w)
            r0, r1 = ref.forward(ids, 0), ref.forward([5], 9)
            qh.put(max(float(np.abs(full[0] - r0).max() / np.abs(r0).max()), float(np.abs(full[1] - r1).max() / np.abs(r1).max())))
    finally:
        dist.destroy_process_group()


def test_tp_plan_hybrid_gdn_reproduces_unsharded_forward():
    """Qwen3.5 hybrid under TP=2: GDN key/value heads, conv channels, gates, [q|gate] rows and out_proj columns
    sharded as csrc/loader.cpp does; 2 all-reduces per layer; prefill + decode (recurrent state stays rank-local)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    qh = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker35, args=(r, 2, port, qh)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert qh.get(timeout=5) < 1e-5


def test_plan_shapes_qwen38_27b_tp8():
    from crane_amd import configs, tp
    c = configs.get_config("qwen3.8-27b")
    for r in range(8):
        p = tp.shard_plan(c, 8, r)
        assert list(p.gdn_key_heads) == [2 * r, 2 * r + 1] and list(p.gdn_value_heads) == list(range(6 * r, 6 * r + 6))
        assert all(v // 3 in p.gdn_key_heads for v in p.gdn_value_heads)      # interleaved pairing stays rank-local
        assert list(p.kv_heads) == [r // 2] and len(p.q_heads) == 3 and len(p.inter) == 2176 and len(p.vocab) == 31040


# ---------------------------------------------------------------------------------------------------------------------------
# The C++ loader's own slicing (csrc/loader.cpp build(), through the host-only cm_tp_shard_plan: the loader's code path over a
# recording source, no device) against crane_amd/tp.py -- the plan the gloo runs above prove -- at the REAL geometries.
# ---------------------------------------------------------------------------------------------------------------------------
COL_SLICED = ("self_attn.o_proj.weight", "mlp.down_proj.weight", "linear_attn.out_proj.weight")


def _expected_ids(cfg, name, shape, plan):
    """source row ids (or column ids for the row-parallel matrices) of `name` that tp.shard_weights hands to this rank, in order"""
    from crane_amd import tp
    t = cfg.get("text_config", cfg)
    if name.endswith(COL_SLICED):
        probe = np.arange(shape[1], dtype=np.int64).reshape(1, -1)
        return tp.shard_weights(t, {name: probe}, plan)[name].reshape(-1)
    probe = np.arange(shape[0], dtype=np.int64).reshape(-1, *([1] * (len(shape) - 1)))
    return tp.shard_weights(t, {name: probe}, plan)[name].reshape(-1)


@pytest.mark.parametrize("name,world", [("qwen3-8b", 8), ("qwen3.8-27b", 8), ("qwen3-0.6b", 4), ("qwen3.5-0.8b", 2), ("tiny-qwen3", 2)])
def test_cpp_loader_slices_what_the_plan_says(name, world):
    """Every rank of TP = 8 on Qwen3-8B (4 q heads + 1 kv head + 1536 MLP columns + 18 992 vocabulary rows per rank) and on
    Qwen3.8-27B (3 q heads, KV head floor(r / 2) replicated on two ranks, 2 GDN key heads + their 6 value heads in HF interleaved
    order, 2176 MLP columns, 31 040 vocabulary rows -- SURVEY 8e): the row / column ranges csrc/loader.cpp copies are exactly
    crane_amd.tp.shard_weights', nothing is copied twice into one shard, and the ranks together cover every tensor."""
    from crane_amd import configs, synth, tp
    from crane_amd.backend import tp_shard_plan
    cfg = configs.get_config(name)
    t = cfg.get("text_config", cfg)
    shapes = {n: s for n, s, *_ in synth.specs_for(cfg)}
    D = t.get("head_dim") or t["hidden_size"] // t["num_attention_heads"]
    hybrid = "linear_num_key_heads" in t
    cover = {}
    for r in range(world):
        plan = tp.shard_plan(t, world, r)
        got = tp_shard_plan(cfg, world, r)
        assert (got["Hq_l"], got["Hkv_l"], got["kvh0"], got["I_l"]) == (len(plan.q_heads), len(plan.kv_heads), plan.kv_heads.start, len(plan.inter))
        assert got["v0"] == (t["vocab_size"] + world - 1) // world * r
        if hybrid:
            assert (got["NK_l"], got["NV_l"]) == (len(plan.gdn_key_heads), len(plan.gdn_value_heads))
        by_tensor = {}
        for c in got["copies"]:
            by_tensor.setdefault(c["tensor"], []).append(c)
        assert set(by_tensor) <= set(shapes), set(by_tensor) - set(shapes)
        for tn, copies in by_tensor.items():
            shape = shapes[tn]
            copies = sorted(copies, key=lambda c: (c["dst"], c["dst_off"]))
            if tn.endswith("conv1d.weight"):          # fetched as one row of conv_dim * k taps: column ranges -> channel ranges
                k = shape[-1]
                ids = np.concatenate([np.arange(c["col0"] // k, (c["col0"] + c["ncols"]) // k) for c in copies])
                assert all(c["col0"] % k == 0 and c["ncols"] % k == 0 for c in copies)
            elif len(shape) == 1:                     # vectors are fetched as [1, n]
                ids = np.concatenate([np.arange(c["col0"], c["col0"] + c["ncols"]) for c in copies])
            elif tn.endswith(COL_SLICED):
                assert all(c["row0"] == 0 and c["nrows"] == shape[0] for c in copies)
                ids = np.concatenate([np.arange(c["col0"], c["col0"] + c["ncols"]) for c in copies])
            else:
                cols = int(np.prod(shape[1:]))
                assert all(c["col0"] == 0 and c["ncols"] == cols for c in copies), tn
                ids = np.concatenate([np.arange(c["row0"], c["row0"] + c["nrows"]) for c in copies])
            assert len(set(ids.tolist())) == len(ids), f"{tn}: an element is copied twice into rank {r}'s shard"
            want = _expected_ids(cfg, tn.replace("model.language_model.", "model."), shape, plan)
            if hybrid and tn.endswith("self_attn.q_proj.weight"):
                # the loader de-interleaves the per-head [q | gate] rows into [all q | all gate]: same rows, another order
                assert sorted(ids.tolist()) == sorted(want.tolist()), tn
                q_rows = [h * 2 * D + d for h in plan.q_heads for d in range(D)]
                assert ids.tolist() == q_rows + [x + D for x in q_rows], tn
            else:
                assert ids.tolist() == want.tolist(), (tn, r)
            cover.setdefault(tn, []).append(ids)
    # the ranks together: every element of a sharded tensor exactly once -- K / V heads tp / Hkv times when Hkv < tp
    for tn, parts in cover.items():
        allids = np.concatenate(parts)
        n = shapes[tn][1] if (tn.endswith(COL_SLICED) and len(shapes[tn]) > 1) else shapes[tn][0]
        counts = np.bincount(allids, minlength=n)
        replicated = not any(tn.endswith(s) for s in ("q_proj.weight", "k_proj.weight", "v_proj.weight", "o_proj.weight", "gate_proj.weight",
                                                      "up_proj.weight", "down_proj.weight", "lm_head.weight", "in_proj_qkv.weight",
                                                      "in_proj_z.weight", "in_proj_b.weight", "in_proj_a.weight", "out_proj.weight",
                                                      "conv1d.weight", "A_log", "dt_bias"))
        if replicated:
            assert (counts == world).all(), tn
        elif tn.endswith(("k_proj.weight", "v_proj.weight")):
            assert (counts == max(1, world // t["num_key_value_heads"])).all(), tn
        elif tn == "lm_head.weight":
            assert (counts[: t["vocab_size"]] == 1).all(), tn
        else:
            assert (counts == 1).all(), tn
