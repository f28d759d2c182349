"""Tensor-parallel partition plan (host logic; the C++ loader `csrc/loader.cpp` + `Model::init_common`
implement exactly this arithmetic).  New design: the reference has no multi-GPU support
(crane-serve/README.md:624).

Column-parallel (output rows split): q/k/v by head, gate/up by intermediate column, lm_head by vocab row.
Row-parallel (input columns split, partial sums all-reduced): o_proj, down_proj.
When num_key_value_heads < tp each KV head is replicated on tp/Hkv consecutive ranks.
"""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class ShardPlan:
    tp: int
    rank: int
    q_heads: range        # global query heads owned by this rank
    kv_heads: range       # global KV heads held by this rank (replicated when Hkv < tp)
    inter: range          # intermediate (MLP) columns
    vocab: range          # lm_head rows
    n_rep: int
    gdn_key_heads: range = range(0)     # Qwen3.5 Gated-Delta-Net key heads owned by this rank
    gdn_value_heads: range = range(0)   # ... and the value heads paired with them (HF interleaved order)

    @property
    def reduces_per_layer(self) -> int:
        return 2 if self.tp > 1 else 0


def shard_plan(cfg: dict, tp: int, rank: int) -> ShardPlan:
    Hq, Hkv = cfg["num_attention_heads"], cfg["num_key_value_heads"]
    I, V = cfg["intermediate_size"], cfg["vocab_size"]
    if not (0 <= rank < tp):
        raise ValueError("rank out of range")
    if Hq % tp or I % tp or (I // tp) % 8:
        raise ValueError("tp must divide attention heads and intermediate size (in multiples of 8)")
    if not (Hkv % tp == 0 or tp % Hkv == 0):
        raise ValueError("tp incompatible with num_key_value_heads")
    hq_l = Hq // tp
    if Hkv >= tp:
        hkv_l, kv0 = Hkv // tp, rank * (Hkv // tp)
    else:
        hkv_l, kv0 = 1, rank * Hkv // tp
    v_l = (V + tp - 1) // tp
    gk, gv = range(0), range(0)
    if "linear_num_key_heads" in cfg:
        NK, NV = cfg["linear_num_key_heads"], cfg["linear_num_value_heads"]
        if NK % tp or NV % tp:
            raise ValueError("tp must divide linear_num_key_heads and linear_num_value_heads")
        gk, gv = range(rank * NK // tp, (rank + 1) * NK // tp), range(rank * NV // tp, (rank + 1) * NV // tp)
    return ShardPlan(tp, rank, range(rank * hq_l, (rank + 1) * hq_l), range(kv0, kv0 + hkv_l),
                     range(rank * (I // tp), (rank + 1) * (I // tp)),
                     range(min(V, rank * v_l), min(V, (rank + 1) * v_l)), hq_l // hkv_l, gk, gv)


def shard_weights(cfg: dict, w: dict, plan: ShardPlan) -> dict:
    """Slice HF-named full tensors into this rank's shard (same row/column ranges the C++ loader copies)."""
    D = cfg.get("head_dim") or cfg["hidden_size"] // cfg["num_attention_heads"]
    qs = slice(plan.q_heads.start * D, plan.q_heads.stop * D)
    ks = slice(plan.kv_heads.start * D, plan.kv_heads.stop * D)
    ins = slice(plan.inter.start, plan.inter.stop)
    out = {}
    hybrid = "linear_num_key_heads" in cfg
    if hybrid:
        K, Vd = cfg["linear_key_head_dim"], cfg["linear_value_head_dim"]
        KDg = cfg["linear_num_key_heads"] * K
        kq = slice(plan.gdn_key_heads.start * K, plan.gdn_key_heads.stop * K)
        vv = slice(plan.gdn_value_heads.start * Vd, plan.gdn_value_heads.stop * Vd)
        nvs = slice(plan.gdn_value_heads.start, plan.gdn_value_heads.stop)
        import numpy as np
    for name, t in w.items():
        if hybrid and name.endswith("self_attn.q_proj.weight"):
            out[name] = t[plan.q_heads.start * 2 * D:plan.q_heads.stop * 2 * D]     # per-head [q | gate] rows
        elif hybrid and (name.endswith("in_proj_qkv.weight") or name.endswith("conv1d.weight")):
            out[name] = np.concatenate([t[kq], t[KDg + kq.start:KDg + kq.stop], t[2 * KDg + vv.start:2 * KDg + vv.stop]], axis=0)
        elif hybrid and name.endswith("in_proj_z.weight"):
            out[name] = t[vv]
        elif hybrid and (name.endswith("in_proj_b.weight") or name.endswith("in_proj_a.weight")
                         or name.endswith("A_log") or name.endswith("dt_bias")):
            out[name] = t[nvs]
        elif hybrid and name.endswith("linear_attn.out_proj.weight"):
            out[name] = t[:, vv]
        elif name.endswith("q_proj.weight"):
            out[name] = t[qs]
        elif name.endswith("k_proj.weight") or name.endswith("v_proj.weight"):
            out[name] = t[ks]
        elif name.endswith("o_proj.weight"):
            out[name] = t[:, qs]
        elif name.endswith("gate_proj.weight") or name.endswith("up_proj.weight"):
            out[name] = t[ins]
        elif name.endswith("down_proj.weight"):
            out[name] = t[:, ins]
        elif name == "lm_head.weight":
            out[name] = t[plan.vocab.start:plan.vocab.stop]
        else:
            out[name] = t                      # embeddings and norms are replicated
    return out
