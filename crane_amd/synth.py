"""Deterministic synthetic checkpoints (host side).

There is no network for real checkpoints, so every benchmark/parity run uses
random-init weights of the named architecture.  To compare a 15 GB model on
the GPU with a CPU oracle without shipping 15 GB around, every tensor element
is a pure function of ``(seed, tensor name, flat index)``:

    h   = fmix32(idx * 0x9E3779B1 + tseed)          (murmur3 finaliser, u32)
    c   = b0 + b1 + b2 + b3 - 510                    (Irwin-Hall(4) ~ normal)
    val = offset + float(c) * mul                    (one f32 rounding)
    w   = bf16_rne(val)

with ``tseed = fmix32(fnv1a32(name) ^ (seed * 0x85EBCA6B + 0x1234567))`` and
``mul = f32(std / sqrt(21845))``.  The HIP generator
(``csrc/synth.hip``), the C oracle (``oracle/c/qwen3_cpu.c``) and this numpy
version produce bit-identical bf16 tensors.

Init convention follows the reference's own ``RandWeights`` test backend
(``crane-core/src/models/qwen3_5/prefill.rs:151-183``): linear ~ N(0, 1/sqrt(fan_in));
norm weights 1 + 0.1*z; embeddings N(0, 1).
"""
from __future__ import annotations

import json
import math
import os
import struct
from typing import Dict, Iterable, List, Tuple

import numpy as np

IH_STD = math.sqrt(21845.0)          # std of b0+b1+b2+b3, bytes uniform on 0..255


def fnv1a32(name: str) -> int:
    h = 0x811C9DC5
    for b in name.encode("utf-8"):
        h = ((h ^ b) * 0x01000193) & 0xFFFFFFFF
    return h


def fmix32_scalar(h: int) -> int:
    h &= 0xFFFFFFFF
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & 0xFFFFFFFF
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & 0xFFFFFFFF
    h ^= h >> 16
    return h


def tensor_seed(name: str, seed: int) -> int:
    return fmix32_scalar(fnv1a32(name) ^ ((seed * 0x85EBCA6B + 0x1234567) & 0xFFFFFFFF))


def _fmix32(h: np.ndarray) -> np.ndarray:
    h ^= h >> np.uint32(16)
    h *= np.uint32(0x85EBCA6B)
    h ^= h >> np.uint32(13)
    h *= np.uint32(0xC2B2AE35)
    h ^= h >> np.uint32(16)
    return h


def synth_mul(std: float) -> np.float32:
    return np.float32(std / IH_STD)


def synth_bf16_bits(name: str, n: int, seed: int, std: float, offset: float = 0.0,
                    start: int = 0) -> np.ndarray:
    """uint16 bf16 bit patterns of elements [start, start+n) of tensor `name`."""
    ts = np.uint32(tensor_seed(name, seed))
    out = np.empty(n, dtype=np.uint16)
    mul = synth_mul(std)
    off = np.float32(offset)
    CH = 1 << 22
    with np.errstate(over="ignore"):
        for s in range(0, n, CH):
            e = min(n, s + CH)
            idx = np.arange(start + s, start + e, dtype=np.uint64).astype(np.uint32)
            h = _fmix32(idx * np.uint32(0x9E3779B1) + ts)
            c = ((h & np.uint32(0xFF)) + ((h >> np.uint32(8)) & np.uint32(0xFF))
                 + ((h >> np.uint32(16)) & np.uint32(0xFF)) + (h >> np.uint32(24))).astype(np.int32) - 510
            val = (off + c.astype(np.float32) * mul).astype(np.float32)
            u = val.view(np.uint32)
            r = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)
            out[s:e] = r.astype(np.uint16)
    return out


def bf16_bits_to_f32(bits: np.ndarray) -> np.ndarray:
    return (bits.astype(np.uint32) << np.uint32(16)).view(np.float32)


def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    return ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)).astype(np.uint16)


# --------------------------------------------------------------------------
# tensor specs: (name, shape, std, offset)   -- HF names, SURVEY Appendix A
# --------------------------------------------------------------------------
Spec = Tuple[str, Tuple[int, ...], float, float]


def qwen3_specs(cfg: dict) -> List[Spec]:
    H = cfg["hidden_size"]
    I = cfg["intermediate_size"]
    Hq = cfg["num_attention_heads"]
    Hkv = cfg["num_key_value_heads"]
    D = cfg.get("head_dim") or H // Hq
    V = cfg["vocab_size"]
    sp: List[Spec] = [("model.embed_tokens.weight", (V, H), 1.0, 0.0)]
    for i in range(cfg["num_hidden_layers"]):
        p = f"model.layers.{i}."
        sp += [
            (p + "self_attn.q_proj.weight", (Hq * D, H), 1 / math.sqrt(H), 0.0),
            (p + "self_attn.k_proj.weight", (Hkv * D, H), 1 / math.sqrt(H), 0.0),
            (p + "self_attn.v_proj.weight", (Hkv * D, H), 1 / math.sqrt(H), 0.0),
            (p + "self_attn.o_proj.weight", (H, Hq * D), 1 / math.sqrt(Hq * D), 0.0),
        ]
        if cfg.get("use_qk_norm", True):
            sp += [(p + "self_attn.q_norm.weight", (D,), 0.1, 1.0),
                   (p + "self_attn.k_norm.weight", (D,), 0.1, 1.0)]
        sp += [
            (p + "mlp.gate_proj.weight", (I, H), 1 / math.sqrt(H), 0.0),
            (p + "mlp.up_proj.weight", (I, H), 1 / math.sqrt(H), 0.0),
            (p + "mlp.down_proj.weight", (H, I), 1 / math.sqrt(I), 0.0),
            (p + "input_layernorm.weight", (H,), 0.1, 1.0),
            (p + "post_attention_layernorm.weight", (H,), 0.1, 1.0),
        ]
    sp.append(("model.norm.weight", (H,), 0.1, 1.0))
    if not cfg.get("tie_word_embeddings", True):
        sp.append(("lm_head.weight", (V, H), 1 / math.sqrt(H), 0.0))
    return sp


def qwen3_5_specs(cfg: dict) -> List[Spec]:
    """Qwen 3.5/3.6/3.8 text model (HF names; shape contract: reference tests/test_qwen35_family_shapes.py:68-131).
    Block / QK norms are stored as w with the model applying (1 + w); the GDN gated norm is a plain weight.
    A_log ~ 0.1 N(0,1) - 2 and linear ~ N(0, 1/sqrt(fan_in)) follow the reference's RandWeights
    (crane-core/src/models/qwen3_5/prefill.rs:151-183)."""
    t = cfg.get("text_config", cfg)
    H, I, V = t["hidden_size"], t["intermediate_size"], t["vocab_size"]
    Hq, Hkv, D = t["num_attention_heads"], t["num_key_value_heads"], t["head_dim"]
    NK, NV = t["linear_num_key_heads"], t["linear_num_value_heads"]
    K, Vd, ker = t["linear_key_head_dim"], t["linear_value_head_dim"], t.get("linear_conv_kernel_dim", 4)
    KD, VD = NK * K, NV * Vd
    interval = t.get("full_attention_interval", 4)
    sH = 1 / math.sqrt(H)
    sp: List[Spec] = [("model.embed_tokens.weight", (V, H), 1.0, 0.0)]
    for i in range(t["num_hidden_layers"]):
        p = f"model.layers.{i}."
        if (i + 1) % interval == 0:
            sp += [(p + "self_attn.q_proj.weight", (Hq * D * 2, H), sH, 0.0),
                   (p + "self_attn.k_proj.weight", (Hkv * D, H), sH, 0.0),
                   (p + "self_attn.v_proj.weight", (Hkv * D, H), sH, 0.0),
                   (p + "self_attn.o_proj.weight", (H, Hq * D), 1 / math.sqrt(Hq * D), 0.0),
                   (p + "self_attn.q_norm.weight", (D,), 0.1, 0.0),
                   (p + "self_attn.k_norm.weight", (D,), 0.1, 0.0)]
        else:
            sp += [(p + "linear_attn.in_proj_qkv.weight", (2 * KD + VD, H), sH, 0.0),
                   (p + "linear_attn.in_proj_z.weight", (VD, H), sH, 0.0),
                   (p + "linear_attn.in_proj_b.weight", (NV, H), sH, 0.0),
                   (p + "linear_attn.in_proj_a.weight", (NV, H), sH, 0.0),
                   (p + "linear_attn.conv1d.weight", (2 * KD + VD, 1, ker), 0.5, 0.0),
                   (p + "linear_attn.A_log", (NV,), 0.1, -2.0),
                   (p + "linear_attn.dt_bias", (NV,), 0.1, 0.0),
                   (p + "linear_attn.norm.weight", (Vd,), 0.1, 1.0),
                   (p + "linear_attn.out_proj.weight", (H, VD), 1 / math.sqrt(VD), 0.0)]
        sp += [(p + "mlp.gate_proj.weight", (I, H), sH, 0.0),
               (p + "mlp.up_proj.weight", (I, H), sH, 0.0),
               (p + "mlp.down_proj.weight", (H, I), 1 / math.sqrt(I), 0.0),
               (p + "input_layernorm.weight", (H,), 0.1, 0.0),
               (p + "post_attention_layernorm.weight", (H,), 0.1, 0.0)]
    sp.append(("model.norm.weight", (H,), 0.1, 0.0))
    if not t.get("tie_word_embeddings", cfg.get("tie_word_embeddings", False)):
        sp.append(("lm_head.weight", (V, H), sH, 0.0))
    return sp


def qwen3_5_vl_specs(cfg: dict) -> List[Spec]:
    """Qwen 3.5-VL: language model under `model.language_model.` (prefix probing, qwen3_5/model.rs:65-74) +
    vision tower under `model.visual.` (qwen3_5/vlm.rs:125; shapes: tests/test_qwen35_family_shapes.py:104-131)."""
    t, v = cfg["text_config"], cfg["vision_config"]
    text = dict(t, tie_word_embeddings=cfg.get("tie_word_embeddings", t.get("tie_word_embeddings", False)))
    dense = str(t.get("model_type", "")).startswith("qwen3_vl")          # Qwen3-VL: the dense Qwen3 decoder as language model
    sp = [((n.replace("model.", "model.language_model.", 1) if n.startswith("model.") else n), s, sd, o)
          for n, s, sd, o in (qwen3_specs(text) if dense else qwen3_5_specs(text))]
    VH, VI, P, T, C = v["hidden_size"], v["intermediate_size"], v["patch_size"], v["temporal_patch_size"], v.get("in_channels", 3)
    M = v.get("spatial_merge_size", 2) ** 2
    p = "model.visual."
    sp += [(p + "patch_embed.proj.weight", (VH, C, T, P, P), 1 / math.sqrt(C * T * P * P), 0.0),
           (p + "patch_embed.proj.bias", (VH,), 0.1, 0.0),
           (p + "pos_embed.weight", (v["num_position_embeddings"], VH), 0.5, 0.0)]
    for i in range(v["depth"]):
        b = f"{p}blocks.{i}."
        sp += [(b + "norm1.weight", (VH,), 0.1, 1.0), (b + "norm1.bias", (VH,), 0.1, 0.0),
               (b + "norm2.weight", (VH,), 0.1, 1.0), (b + "norm2.bias", (VH,), 0.1, 0.0),
               (b + "attn.qkv.weight", (3 * VH, VH), 1 / math.sqrt(VH), 0.0), (b + "attn.qkv.bias", (3 * VH,), 0.1, 0.0),
               (b + "attn.proj.weight", (VH, VH), 1 / math.sqrt(VH), 0.0), (b + "attn.proj.bias", (VH,), 0.1, 0.0),
               (b + "mlp.linear_fc1.weight", (VI, VH), 1 / math.sqrt(VH), 0.0), (b + "mlp.linear_fc1.bias", (VI,), 0.1, 0.0),
               (b + "mlp.linear_fc2.weight", (VH, VI), 1 / math.sqrt(VI), 0.0), (b + "mlp.linear_fc2.bias", (VH,), 0.1, 0.0)]
    m = p + "merger."
    sp += [(m + "norm.weight", (VH,), 0.1, 1.0), (m + "norm.bias", (VH,), 0.1, 0.0),
           (m + "linear_fc1.weight", (VH * M, VH * M), 1 / math.sqrt(VH * M), 0.0), (m + "linear_fc1.bias", (VH * M,), 0.1, 0.0),
           (m + "linear_fc2.weight", (v["out_hidden_size"], VH * M), 1 / math.sqrt(VH * M), 0.0),
           (m + "linear_fc2.bias", (v["out_hidden_size"],), 0.1, 0.0)]
    for k in range(len(v.get("deepstack_visual_indexes", []))):          # DeepStack mergers: LayerNorm over the regrouped 4 x hidden row
        m = f"{p}deepstack_merger_list.{k}."
        sp += [(m + "norm.weight", (VH * M,), 0.1, 1.0), (m + "norm.bias", (VH * M,), 0.1, 0.0),
               (m + "linear_fc1.weight", (VH * M, VH * M), 1 / math.sqrt(VH * M), 0.0), (m + "linear_fc1.bias", (VH * M,), 0.1, 0.0),
               (m + "linear_fc2.weight", (v["out_hidden_size"], VH * M), 1 / math.sqrt(VH * M), 0.0),
               (m + "linear_fc2.bias", (v["out_hidden_size"],), 0.1, 0.0)]
    return sp


def specs_for(cfg: dict) -> List[Spec]:
    mt = cfg.get("model_type", "qwen3")
    if "vision_config" in cfg and "text_config" in cfg:
        return qwen3_5_vl_specs(cfg)
    if mt == "qwen3":
        return qwen3_specs(cfg)
    if mt in ("qwen3_5", "qwen3_5_text"):
        return qwen3_5_specs(cfg)
    raise ValueError(f"no synthetic spec for model_type {mt!r}")


def synth_weights_f32(cfg: dict, seed: int = 0) -> Dict[str, np.ndarray]:
    """bf16-rounded synthetic weights up-cast to f32 (what the oracle consumes)."""
    out = {}
    for name, shape, std, off in specs_for(cfg):
        n = int(np.prod(shape))
        out[name] = bf16_bits_to_f32(synth_bf16_bits(name, n, seed, std, off)).reshape(shape)
    return out


# --------------------------------------------------------------------------
# safetensors writer (BF16), exact HF layout: u64 header_len | JSON | data
# --------------------------------------------------------------------------
def write_safetensors_bf16(path: str, tensors: Iterable[Tuple[str, Tuple[int, ...], np.ndarray]]):
    """tensors: iterable of (name, shape, uint16 bf16 bits flat)."""
    items = list(tensors)
    header, off = {}, 0
    for name, shape, bits in items:
        nb = int(bits.size) * 2
        header[name] = {"dtype": "BF16", "shape": list(shape), "data_offsets": [off, off + nb]}
        off += nb
    hj = json.dumps(header, separators=(",", ":")).encode()
    hj += b" " * ((8 - len(hj) % 8) % 8)
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(hj)))
        f.write(hj)
        for _, _, bits in items:
            f.write(np.ascontiguousarray(bits, dtype=np.uint16).tobytes())


def write_model_dir(path: str, cfg: dict, seed: int = 0, shards: int = 1) -> str:
    """Write config.json + (sharded) model safetensors like an HF checkpoint dir
    (shard discovery contract: crane-core/src/utils/utils.rs:16-57)."""
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(cfg, f, indent=1)
    specs = specs_for(cfg)
    if shards <= 1:
        write_safetensors_bf16(
            os.path.join(path, "model.safetensors"),
            ((n, s, synth_bf16_bits(n, int(np.prod(s)), seed, std, off)) for n, s, std, off in specs))
    else:
        per = (len(specs) + shards - 1) // shards
        weight_map = {}
        for k in range(shards):
            fn = f"model-{k + 1:05d}-of-{shards:05d}.safetensors"
            part = specs[k * per:(k + 1) * per]
            write_safetensors_bf16(
                os.path.join(path, fn),
                ((n, s, synth_bf16_bits(n, int(np.prod(s)), seed, std, off)) for n, s, std, off in part))
            for n, *_ in part:
                weight_map[n] = fn
        with open(os.path.join(path, "model.safetensors.index.json"), "w") as f:
            json.dump({"metadata": {}, "weight_map": weight_map}, f)
    return path
