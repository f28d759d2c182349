"""Named model configurations (HF ``config.json`` dictionaries).

Shapes from SURVEY.md section 8 (values tagged [external] there are the
HF-published ones).  ``tiny*`` configs are CPU-oracle-sized test models; the
first mirrors the reference's own unit-test config
(``crane-core/src/models/qwen3/modeling.rs:1386-1403``) scaled to dims the HIP
kernels accept (hidden/head_dim multiples of 64).
"""
from __future__ import annotations

import copy

_QWEN3_COMMON = dict(
    model_type="qwen3", rms_norm_eps=1e-6, rope_theta=1_000_000.0, attention_bias=False,
    use_qk_norm=True, max_position_embeddings=40960, torch_dtype="bfloat16",
)

CONFIGS = {
    "qwen3-0.6b": dict(_QWEN3_COMMON, vocab_size=151936, hidden_size=1024, intermediate_size=3072,
                       num_hidden_layers=28, num_attention_heads=16, num_key_value_heads=8,
                       head_dim=128, tie_word_embeddings=True),
    "qwen3-8b": dict(_QWEN3_COMMON, vocab_size=151936, hidden_size=4096, intermediate_size=12288,
                     num_hidden_layers=36, num_attention_heads=32, num_key_value_heads=8,
                     head_dim=128, tie_word_embeddings=False),
    # the headline geometry (every projection shape, GQA group and the 151 936-row lm_head of Qwen3-8B) at a depth the
    # CPU oracle finishes in seconds: the parity tests of the benchmarked kernel instantiations
    "qwen3-8b-2l": dict(_QWEN3_COMMON, vocab_size=151936, hidden_size=4096, intermediate_size=12288,
                        num_hidden_layers=2, num_attention_heads=32, num_key_value_heads=8,
                        head_dim=128, tie_word_embeddings=False),
    "qwen3-0.6b-2l": dict(_QWEN3_COMMON, vocab_size=151936, hidden_size=1024, intermediate_size=3072,
                          num_hidden_layers=2, num_attention_heads=16, num_key_value_heads=8,
                          head_dim=128, tie_word_embeddings=True),
    # widths that are multiples of 2048 (what the persistent chain kernel needs), small enough for the numpy oracle
    "eng-qwen3": dict(_QWEN3_COMMON, vocab_size=2048, hidden_size=2048, intermediate_size=4096,
                      num_hidden_layers=3, num_attention_heads=32, num_key_value_heads=8,
                      head_dim=128, tie_word_embeddings=False, max_position_embeddings=8192),
    # the same with a GQA group of 2 (16 q / 8 kv heads: the text decoder of Qwen3-VL-2B, Qwen3-1.7B)
    "eng-qwen3-gqa2": dict(_QWEN3_COMMON, vocab_size=2048, hidden_size=2048, intermediate_size=6144,
                           num_hidden_layers=3, num_attention_heads=16, num_key_value_heads=8,
                           head_dim=128, tie_word_embeddings=True, max_position_embeddings=8192),
    # Qwen3-0.6B widths (hidden 1024, intermediate 3072: multiples of 1024 only -> the persistent kernel's 1024-element chunks)
    "eng-qwen3-h1024": dict(_QWEN3_COMMON, vocab_size=2048, hidden_size=1024, intermediate_size=3072,
                            num_hidden_layers=3, num_attention_heads=16, num_key_value_heads=8,
                            head_dim=128, tie_word_embeddings=True, max_position_embeddings=8192),
    # small shapes for CPU-oracle parity
    "tiny-qwen3": dict(_QWEN3_COMMON, vocab_size=512, hidden_size=256, intermediate_size=512,
                       num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                       head_dim=128, tie_word_embeddings=True, max_position_embeddings=4096),
    "tiny-qwen3-untied": dict(_QWEN3_COMMON, vocab_size=1000, hidden_size=512, intermediate_size=1536,
                              num_hidden_layers=3, num_attention_heads=8, num_key_value_heads=2,
                              head_dim=128, tie_word_embeddings=False, max_position_embeddings=4096),
    "small-qwen3": dict(_QWEN3_COMMON, vocab_size=8192, hidden_size=1024, intermediate_size=3072,
                        num_hidden_layers=4, num_attention_heads=16, num_key_value_heads=8,
                        head_dim=128, tie_word_embeddings=True, max_position_embeddings=8192),
}


_QWEN35_COMMON = dict(
    model_type="qwen3_5_text", rms_norm_eps=1e-6, attention_bias=False, hidden_act="silu",
    linear_conv_kernel_dim=4, full_attention_interval=4, attn_output_gate=True, torch_dtype="bfloat16",
    rope_parameters=dict(rope_type="default", rope_theta=10_000_000.0, partial_rotary_factor=0.25,
                         mrope_section=[11, 11, 10], mrope_interleaved=True),
)
CONFIGS.update({
    # Qwen3.5-0.8B (values tagged [external] there)
    "qwen3.5-0.8b": dict(_QWEN35_COMMON, vocab_size=248320, hidden_size=1024, intermediate_size=3584,
                         num_hidden_layers=24, num_attention_heads=8, num_key_value_heads=2, head_dim=256,
                         linear_key_head_dim=128, linear_value_head_dim=128, linear_num_key_heads=16,
                         linear_num_value_heads=16, tie_word_embeddings=True, max_position_embeddings=262144),
    # Qwen3.5-2B: the model of the reference's only published numbers for this path (README.md:86-87,496-503: Qwen3.5-2B-Q8_0 on an
    # RX 7800 XT).  The reference does not spell its geometry out; this is the HF checkpoint's config as far as it is known here
    # (hidden 2048, 24 layers, 8 q / 2 kv heads of 256, 16 + 16 linear heads of 128): 1.88 B parameters
    "qwen3.5-2b": dict(_QWEN35_COMMON, vocab_size=248320, hidden_size=2048, intermediate_size=6144,
                       num_hidden_layers=24, num_attention_heads=8, num_key_value_heads=2, head_dim=256,
                       linear_key_head_dim=128, linear_value_head_dim=128, linear_num_key_heads=16,
                       linear_num_value_heads=16, tie_word_embeddings=True, max_position_embeddings=262144),
    # Qwen3.8-27B: fully pinned by the reference (qwen3_5/config.rs:298-324,332-364)
    "qwen3.8-27b": dict(_QWEN35_COMMON, vocab_size=248320, hidden_size=5120, intermediate_size=17408,
                        num_hidden_layers=64, num_attention_heads=24, num_key_value_heads=4, head_dim=256,
                        linear_key_head_dim=128, linear_value_head_dim=128, linear_num_key_heads=16,
                        linear_num_value_heads=48, tie_word_embeddings=False, max_position_embeddings=262144),
    # CPU-oracle sized hybrid: 3 GDN + 1 full layer, twice; value heads = 2 x key heads (interleaved pairing)
    "tiny-qwen3.5": dict(_QWEN35_COMMON, vocab_size=512, hidden_size=256, intermediate_size=512,
                         num_hidden_layers=8, num_attention_heads=4, num_key_value_heads=2, head_dim=256,
                         linear_key_head_dim=128, linear_value_head_dim=128, linear_num_key_heads=2,
                         linear_num_value_heads=4, tie_word_embeddings=False, max_position_embeddings=4096),
})


def _vl(text: dict, vision: dict, image_token_id: int) -> dict:
    t = dict(text)
    tie = t.pop("tie_word_embeddings", False)
    return dict(model_type="qwen3_5", text_config=t, vision_config=vision, tie_word_embeddings=tie,
                image_token_id=image_token_id, vision_start_token_id=image_token_id - 1,
                vision_end_token_id=image_token_id + 1, torch_dtype="bfloat16")


_VISION_COMMON = dict(model_type="qwen3_5_vision", hidden_act="gelu_pytorch_tanh", in_channels=3, patch_size=16,
                      temporal_patch_size=2, spatial_merge_size=2)
CONFIGS.update({
    # Qwen3-VL-2B-class tower ([external]): depth 24, hidden 1024, 16 heads, inter 4096, pos-emb 2304;
    # text side = the reference's LIVE VL text model (qwen3_5/vlm.rs): Qwen3.5-0.8B
    "qwen3.5-vl-0.8b": _vl(CONFIGS["qwen3.5-0.8b"],
                           dict(_VISION_COMMON, depth=24, hidden_size=1024, num_heads=16, intermediate_size=4096,
                                out_hidden_size=1024, num_position_embeddings=2304), 248056),
    "tiny-qwen3.5-vl": _vl(CONFIGS["tiny-qwen3.5"],
                           dict(_VISION_COMMON, depth=2, hidden_size=256, num_heads=4, intermediate_size=512,
                                out_hidden_size=256, num_position_embeddings=64), 500),
})


# Qwen3-VL (BASELINE configs[3]): dense Qwen3 text model with 3-axis interleaved MRoPE + the same vision tower plus DeepStack
# mergers after the listed blocks (reference: crane-core/src/models/qwen3_vl/; HF Qwen3VLConfig)
_QWEN3VL_TEXT = dict(model_type="qwen3_vl_text", rms_norm_eps=1e-6, attention_bias=False, hidden_act="silu", torch_dtype="bfloat16",
                     rope_parameters=dict(rope_type="default", rope_theta=5_000_000.0, mrope_section=[24, 20, 20], mrope_interleaved=True))


def _vl3(text: dict, vision: dict, image_token_id: int) -> dict:
    t = dict(_QWEN3VL_TEXT, **text)
    tie = t.pop("tie_word_embeddings", True)
    return dict(model_type="qwen3_vl", text_config=t, vision_config=dict(vision, model_type="qwen3_vl"), tie_word_embeddings=tie,
                image_token_id=image_token_id, video_token_id=image_token_id + 1, vision_start_token_id=image_token_id - 3,
                vision_end_token_id=image_token_id - 2, torch_dtype="bfloat16")


CONFIGS.update({
    # Qwen3-VL-2B ([external]): text 2048 / 28 layers / 16 q / 8 kv / 128 / 6144 / 151 936 tied;
    # vision depth 24, hidden 1024, 16 heads, inter 4096, patch 16, merge 2, out 2048, pos-emb 2304, deepstack [5, 11, 17]
    "qwen3-vl-2b": _vl3(dict(vocab_size=151936, hidden_size=2048, intermediate_size=6144, num_hidden_layers=28,
                             num_attention_heads=16, num_key_value_heads=8, head_dim=128, tie_word_embeddings=True,
                             max_position_embeddings=262144),
                        dict(_VISION_COMMON, depth=24, hidden_size=1024, num_heads=16, intermediate_size=4096, out_hidden_size=2048,
                             num_position_embeddings=2304, deepstack_visual_indexes=[5, 11, 17]), 151655),
    "tiny-qwen3-vl": _vl3(dict(vocab_size=512, hidden_size=256, intermediate_size=512, num_hidden_layers=3,
                               num_attention_heads=4, num_key_value_heads=2, head_dim=128, tie_word_embeddings=False,
                               max_position_embeddings=4096),
                          dict(_VISION_COMMON, depth=3, hidden_size=256, num_heads=4, intermediate_size=512, out_hidden_size=256,
                               num_position_embeddings=64, deepstack_visual_indexes=[0, 1]), 500),
})


def get_config(name: str) -> dict:
    if name not in CONFIGS:
        raise KeyError(f"unknown config {name!r}; have {sorted(CONFIGS)}")
    return copy.deepcopy(CONFIGS[name])


def synthetic_prompt(n: int, vocab: int):
    """ids[i] = (7 i + 3) mod V  -- the pattern of qwen3_5/prefill.rs:249."""
    return [(7 * i + 3) % vocab for i in range(n)]
