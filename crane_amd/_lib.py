"""ctypes binding of include/crane_mi355.h.

The HIP library is the product: importing a model class without it raises
(there is no CPU fallback, by design).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcrane_mi355.so")

CM_ABI_VERSION = 3
CM_OK = 0
STATUS_NAMES = {0: "CM_OK", -1: "CM_ERR_INVALID", -2: "CM_ERR_IO", -3: "CM_ERR_UNSUPPORTED",
                -4: "CM_ERR_DEVICE", -5: "CM_ERR_OOM", -6: "CM_ERR_RANGE"}

# every symbol include/crane_mi355.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "cm_create", "cm_create_synthetic", "cm_destroy", "cm_last_error", "cm_last_global_error",
    "cm_tp_unique_id", "cm_num_layers", "cm_vocab_size", "cm_hidden_size", "cm_max_seq_len",
    "cm_kv_bytes", "cm_weight_bytes", "cm_decode_bytes_per_token", "cm_tp_ranks", "cm_engine_active", "cm_forward_step",
    "cm_forward_step_greedy", "cm_clear_kv", "cm_warmup", "cm_generate", "cm_seq_alloc",
    "cm_seq_free", "cm_seq_fork", "cm_seq_len", "cm_seq_truncate", "cm_seq_forward",
    "cm_decode_batch", "cm_prefill_batch", "cm_image_smart_resize", "cm_image_preprocess", "cm_preprocess_last_error", "cm_image_token_id", "cm_vision_encode", "cm_vlm_forward", "cm_embed_tokens", "cm_forward_embeds", "cm_sample", "cm_topk", "cm_read_logits", "cm_engine_create", "cm_engine_destroy", "cm_engine_submit", "cm_engine_cancel",
    "cm_gguf_config", "cm_checkpoint_inspect", "cm_tp_shard_plan", "cm_engine_step", "cm_engine_step_many", "cm_engine_has_work", "cm_engine_get_stats", "cm_engine_last_error", "cm_bench_decode", "cm_bench_kernel", "cm_debug_fill_kv", "cm_debug_read", "cm_debug_qgemv", "cm_debug_qgemm", "cm_debug_set", "cm_debug_peer_selftest", "cm_debug_qgemm_plan",
]


class CmOpts(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("device", C.c_int32), ("tp_rank", C.c_int32),
        ("tp_size", C.c_int32), ("tp_unique_id", C.c_void_p), ("max_seq_len", C.c_uint32),
        ("max_seqs", C.c_uint32), ("kv_block_size", C.c_uint32), ("kv_pool_tokens", C.c_uint64),
        ("kv_dtype", C.c_int32), ("use_graph", C.c_int32), ("prefill_chunk", C.c_uint32),
        ("prefill_split", C.c_int32), ("isq", C.c_uint32), ("engine", C.c_int32), ("debug_flags", C.c_uint32),
        ("tp_mode", C.c_uint32), ("tp_devices", C.POINTER(C.c_int32)), ("tp_collective", C.c_uint32),
        ("reserved", C.c_uint32 * 1),
    ]


assert C.sizeof(CmOpts) == 96          # the round-4 fields live in the former reserved words (include/crane_mi355.h)


class CmPreprocConfig(C.Structure):
    _fields_ = [("patch_size", C.c_uint32), ("temporal_patch_size", C.c_uint32), ("merge_size", C.c_uint32), ("reserved", C.c_uint32),
                ("min_pixels", C.c_uint64), ("max_pixels", C.c_uint64), ("image_mean", C.c_float * 3), ("image_std", C.c_float * 3)]


class CmGenConfig(C.Structure):
    _fields_ = [
        ("max_new_tokens", C.c_uint32), ("temperature", C.c_float), ("top_p", C.c_float),
        ("repetition_penalty", C.c_float), ("repeat_last_n", C.c_uint32),
        ("eos_token_id", C.c_int64 * 4), ("sync_every", C.c_uint32), ("top_k", C.c_uint32), ("seed_lo", C.c_uint32), ("seed_hi", C.c_uint32),
        ("frequency_penalty", C.c_float), ("presence_penalty", C.c_float), ("reserved", C.c_uint32 * 2),
    ]


class CmSampleParams(C.Structure):
    _fields_ = [
        ("temperature", C.c_float), ("top_p", C.c_float), ("top_k", C.c_uint32), ("repetition_penalty", C.c_float),
        ("frequency_penalty", C.c_float), ("presence_penalty", C.c_float), ("repeat_last_n", C.c_uint32),
        ("draw", C.c_uint32), ("seed", C.c_uint64), ("reserved", C.c_uint32 * 6),
    ]


class CmEngineOpts(C.Structure):
    _fields_ = [("max_running", C.c_uint32), ("repeat_last_n", C.c_uint32), ("seed", C.c_uint64), ("batch_prefill", C.c_int32),
                ("reserved", C.c_uint32 * 7)]


class CmRequest(C.Structure):
    _fields_ = [
        ("tokens", C.POINTER(C.c_uint32)), ("n_tokens", C.c_size_t), ("max_tokens", C.c_uint32), ("temperature", C.c_float),
        ("top_p", C.c_float), ("top_k", C.c_uint32), ("repetition_penalty", C.c_float), ("frequency_penalty", C.c_float),
        ("presence_penalty", C.c_float), ("eos_token_id", C.c_int64 * 4), ("seed", C.c_uint64), ("reserved", C.c_uint32 * 4),
    ]


class CmEngineEvent(C.Structure):
    _fields_ = [
        ("req_id", C.c_uint64), ("kind", C.c_uint32), ("token", C.c_uint32), ("finish_reason", C.c_uint32),
        ("prompt_tokens", C.c_uint32), ("completion_tokens", C.c_uint32), ("error", C.c_int32),
    ]


class CmEngineStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("waiting", "running", "completed", "failed", "preemptions", "prompt_tokens",
                                          "completion_tokens", "prefill_steps", "decode_rounds", "free_pages", "total_pages")] + \
               [("reserved", C.c_uint64 * 5)]


TOKEN_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32)


class CraneError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"{STATUS_NAMES.get(code, code)}: {msg}")
        self.code = code


_lib = None


def load():
    """Load libcrane_mi355.so (fails loudly when it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP library first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C crane_amd/csrc). "
            "There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    P = C.POINTER
    vp, u32p, f32p = C.c_void_p, P(C.c_uint32), P(C.c_float)
    lib.cm_create.argtypes = [C.c_char_p, P(CmOpts), P(vp)]
    lib.cm_create_synthetic.argtypes = [C.c_char_p, C.c_uint64, P(CmOpts), P(vp)]
    lib.cm_destroy.argtypes = [vp]
    lib.cm_destroy.restype = None
    lib.cm_last_error.argtypes = [vp]
    lib.cm_last_error.restype = C.c_char_p
    lib.cm_last_global_error.restype = C.c_char_p
    lib.cm_tp_unique_id.argtypes = [vp]
    for n in ("cm_num_layers", "cm_vocab_size", "cm_hidden_size", "cm_max_seq_len"):
        getattr(lib, n).argtypes = [vp]
        getattr(lib, n).restype = C.c_size_t
    for n in ("cm_kv_bytes", "cm_weight_bytes"):
        getattr(lib, n).argtypes = [vp]
        getattr(lib, n).restype = C.c_uint64
    lib.cm_tp_ranks.argtypes = [vp]
    lib.cm_engine_active.argtypes = [vp]
    lib.cm_decode_bytes_per_token.argtypes = [vp, C.c_size_t]
    lib.cm_decode_bytes_per_token.restype = C.c_uint64
    lib.cm_forward_step.argtypes = [vp, u32p, C.c_size_t, C.c_size_t, f32p]
    lib.cm_forward_step_greedy.argtypes = [vp, u32p, C.c_size_t, C.c_size_t, u32p]
    lib.cm_clear_kv.argtypes = [vp]
    lib.cm_clear_kv.restype = None
    lib.cm_warmup.argtypes = [vp]
    lib.cm_generate.argtypes = [vp, u32p, C.c_size_t, P(CmGenConfig), u32p, P(C.c_size_t), TOKEN_CB, vp]
    lib.cm_seq_alloc.argtypes = [vp, P(C.c_int32)]
    lib.cm_seq_free.argtypes = [vp, C.c_int32]
    lib.cm_seq_fork.argtypes = [vp, C.c_int32, P(C.c_int32)]
    lib.cm_seq_len.argtypes = [vp, C.c_int32]
    lib.cm_seq_len.restype = C.c_int64
    lib.cm_debug_peer_selftest.restype = C.c_long
    lib.cm_debug_peer_selftest.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    lib.cm_debug_qgemm_plan.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.c_int32, P(C.c_int64)]
    lib.cm_seq_truncate.argtypes = [vp, C.c_int32, C.c_size_t]
    lib.cm_seq_forward.argtypes = [vp, C.c_int32, u32p, C.c_size_t, C.c_size_t, f32p, u32p]
    lib.cm_decode_batch.argtypes = [vp, P(C.c_int32), u32p, C.c_size_t, f32p, u32p]
    lib.cm_prefill_batch.argtypes = [vp, P(C.c_int32), P(P(C.c_uint32)), P(C.c_size_t), C.c_size_t, f32p, u32p]
    lib.cm_image_smart_resize.argtypes = [P(CmPreprocConfig), C.c_uint32, C.c_uint32, u32p, u32p]
    lib.cm_image_preprocess.argtypes = [P(CmPreprocConfig), C.c_char_p, C.c_uint32, C.c_uint32, f32p, C.c_size_t, u32p, P(C.c_size_t)]
    lib.cm_preprocess_last_error.restype = C.c_char_p
    lib.cm_image_token_id.argtypes = [vp]
    lib.cm_image_token_id.restype = C.c_int64
    lib.cm_vision_encode.argtypes = [vp, f32p, C.c_size_t, u32p, C.c_size_t, f32p, P(C.c_size_t)]
    lib.cm_vlm_forward.argtypes = [vp, C.c_int32, u32p, C.c_size_t, C.c_size_t, f32p, C.c_size_t, u32p, C.c_size_t, f32p, u32p]
    lib.cm_embed_tokens.argtypes = [vp, u32p, C.c_size_t, f32p]
    lib.cm_forward_embeds.argtypes = [vp, C.c_int32, f32p, C.c_size_t, P(C.c_int32), C.c_size_t, f32p, u32p]
    lib.cm_sample.argtypes = [vp, P(CmSampleParams), u32p, C.c_size_t, u32p]
    lib.cm_topk.argtypes = [vp, f32p, C.c_size_t, C.c_uint32, u32p, f32p]
    lib.cm_read_logits.argtypes = [vp, f32p]
    lib.cm_engine_create.argtypes = [vp, P(CmEngineOpts), P(vp)]
    lib.cm_engine_destroy.argtypes = [vp]
    lib.cm_engine_destroy.restype = None
    lib.cm_engine_submit.argtypes = [vp, P(CmRequest), P(C.c_uint64)]
    lib.cm_engine_cancel.argtypes = [vp, C.c_uint64]
    lib.cm_gguf_config.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, P(C.c_size_t)]
    lib.cm_checkpoint_inspect.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, P(C.c_size_t)]
    lib.cm_tp_shard_plan.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_char_p, C.c_size_t, P(C.c_size_t)]
    lib.cm_engine_step.argtypes = [vp, P(CmEngineEvent), C.c_size_t, P(C.c_size_t)]
    lib.cm_engine_step_many.argtypes = [vp, C.c_size_t, P(CmEngineEvent), C.c_size_t, P(C.c_size_t)]
    lib.cm_engine_has_work.argtypes = [vp]
    lib.cm_engine_get_stats.argtypes = [vp, P(CmEngineStats)]
    lib.cm_engine_last_error.argtypes = [vp]
    lib.cm_engine_last_error.restype = C.c_char_p
    lib.cm_bench_decode.argtypes = [vp, C.c_uint32, C.c_size_t, u32p, f32p]
    lib.cm_bench_kernel.argtypes = [vp, C.c_char_p, C.c_size_t, f32p, P(C.c_uint64)]
    lib.cm_debug_fill_kv.argtypes = [vp, C.c_size_t, C.c_uint64]
    lib.cm_debug_read.argtypes = [vp, C.c_char_p, f32p, C.c_size_t]
    lib.cm_debug_qgemv.argtypes = [vp, C.c_int32, C.c_char_p, f32p, C.c_size_t, f32p, C.c_size_t]
    lib.cm_debug_qgemm.argtypes = [vp, C.c_int32, C.c_char_p, f32p, C.c_size_t, C.c_size_t, f32p, C.c_size_t]
    lib.cm_debug_set.argtypes = [vp, C.c_char_p, C.c_int64]
    _lib = lib
    return lib
