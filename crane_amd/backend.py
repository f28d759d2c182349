"""Python mirror of the reference's model-side interface, on top of the C ABI.

Method names, argument meaning and error behaviour follow

* ``trait ModelBackend``      crane-serve/src/engine/backend.rs:30-147
* ``trait ModelForCausalLM``  crane-core/src/generation/based.rs:5-34
* ``struct GenerationConfig`` crane-core/src/generation/mod.rs:62-99
* ``qwen3::Model``            crane-core/src/models/qwen3/model.rs:45-349

so the parity tests read like the reference's own.  Host Rust is not
available in this image; INTEGRATION.md has the Rust shim that binds the same
C ABI.  All compute happens in ``libcrane_mi355.so`` (HIP); this module never
touches the oracle and has no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import json
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence

import numpy as np

from . import _lib


@dataclass
class GenerationConfig:
    """generation/mod.rs:62-99 (same field names and defaults)."""
    max_new_tokens: int = 245
    temperature: Optional[float] = 0.67
    top_p: Optional[float] = 1.0
    repetition_penalty: float = 1.0
    repeat_last_n: int = 5
    do_sample: bool = False          # never read by the reference either
    pad_token_id: Optional[int] = None
    eos_token_id: Optional[int] = None
    report_speed: bool = False
    enable_thinking: Optional[bool] = None
    # sampler knobs of the serving engine (crane-serve/src/engine/sampling.rs SamplingParams), not in generation/mod.rs
    top_k: int = 0
    frequency_penalty: float = 0.0
    presence_penalty: float = 0.0
    seed: int = 299792458

    @classmethod
    def with_max_tokens(cls, n: int) -> "GenerationConfig":
        return cls(max_new_tokens=n)

    @classmethod
    def greedy(cls, n: int, eos_token_id: Optional[int] = None) -> "GenerationConfig":
        """temperature: None selects arg-max (qwen3/model.rs:284)."""
        return cls(max_new_tokens=n, temperature=None, top_p=None, eos_token_id=eos_token_id)


class TokenStreamer:
    """generation/streamer.rs:7-10."""

    def append(self, token_id: int) -> None:  # pragma: no cover - interface
        raise NotImplementedError

    def finalize(self) -> None:  # pragma: no cover - interface
        pass


class ListStreamer(TokenStreamer):
    def __init__(self):
        self.tokens: List[int] = []
        self.finalized = False

    def append(self, token_id: int) -> None:
        self.tokens.append(int(token_id))

    def finalize(self) -> None:
        self.finalized = True


def _u32(ids: Sequence[int]):
    arr = np.ascontiguousarray(np.asarray(ids, dtype=np.uint32))
    return arr, arr.ctypes.data_as(C.POINTER(C.c_uint32))


def _host_json(fn_name: str, path: str) -> dict:
    import json
    lib = _lib.load()
    fn = getattr(lib, fn_name)
    need = C.c_size_t(0)
    rc = fn(path.encode(), None, 0, C.byref(need))
    if rc != 0:
        raise _lib.CraneError(rc, lib.cm_last_global_error().decode())
    buf = C.create_string_buffer(need.value)
    rc = fn(path.encode(), buf, need.value, C.byref(need))
    if rc != 0:
        raise _lib.CraneError(rc, lib.cm_last_global_error().decode())
    return json.loads(buf.value.decode())


def gguf_config(path: str) -> dict:
    """config.json as the loader derives it from a GGUF file (cm_gguf_config; host only, no GPU needed)."""
    return _host_json("cm_gguf_config", path)


def tp_shard_plan(cfg: dict, tp_size: int, tp_rank: int) -> dict:
    """The copies the C++ loader makes for one tensor-parallel rank + the rank's geometry (cm_tp_shard_plan; host only)."""
    import json
    lib = _lib.load()
    txt = json.dumps(cfg).encode()
    need = C.c_size_t(0)
    rc = lib.cm_tp_shard_plan(txt, tp_size, tp_rank, None, 0, C.byref(need))
    if rc != 0:
        raise _lib.CraneError(rc, lib.cm_last_global_error().decode())
    buf = C.create_string_buffer(need.value)
    rc = lib.cm_tp_shard_plan(txt, tp_size, tp_rank, buf, need.value, C.byref(need))
    if rc != 0:
        raise _lib.CraneError(rc, lib.cm_last_global_error().decode())
    return json.loads(buf.value.decode())


def checkpoint_inspect(model_dir: str) -> dict:
    """Tensor directory of a (sharded) safetensors checkpoint as the C++ loader sees it (cm_checkpoint_inspect; host only)."""
    return _host_json("cm_checkpoint_inspect", model_dir)


class Model:
    """One replica / TP rank of the MI355X inference path (qwen3::Model + ModelBackend)."""

    def __init__(self, handle, lib):
        self._h = handle
        self._lib = lib
        self.vocab_size = int(lib.cm_vocab_size(handle))
        self.hidden_size = int(lib.cm_hidden_size(handle))
        self._eos: List[int] = []

    # -- construction -------------------------------------------------------
    @staticmethod
    def _opts(device=0, max_seq_len=0, max_seqs=0, kv_block_size=0, kv_pool_tokens=0, use_graph=0,
              tp_rank=0, tp_size=1, tp_unique_id: Optional[bytes] = None, prefill_chunk=0, prefill_split=0,
              kv_dtype="f16", isq: Optional[str] = None, engine=0, debug_tp_local=False, debug_force_rccl=False,
              tp_in_process=False, tp_devices: Optional[List[int]] = None, tp_collective: Optional[str] = None):
        o = _lib.CmOpts()
        o.abi_version = _lib.CM_ABI_VERSION
        o.device, o.tp_rank, o.tp_size = device, tp_rank, tp_size
        o.max_seq_len, o.max_seqs, o.kv_block_size = max_seq_len, max_seqs, kv_block_size
        o.kv_pool_tokens, o.use_graph = kv_pool_tokens, use_graph
        o.prefill_chunk, o.prefill_split = prefill_chunk, prefill_split
        o.kv_dtype = {"f16": 0, "f32": 1, "int8": 2, "int4": 3, "bf16": 4}[kv_dtype]
        o.isq = {None: 0, "none": 0, "q8_0": 8, "q4_0": 2, "q5_0": 6}[isq.lower() if isinstance(isq, str) else isq]     # --quant / CRANE_ISQ
        o.engine = int(engine)                                    # persistent chain kernel: 0 default, 1 require, -1 off
        o.debug_flags = (1 if debug_tp_local else 0) | (2 if debug_force_rccl else 0)
        keep = None
        if tp_unique_id is not None:
            keep = C.create_string_buffer(bytes(tp_unique_id), 128)
            o.tp_unique_id = C.cast(keep, C.c_void_p)
        if tp_in_process:                # ONE handle owns all tp_size ranks (CM_TP_IN_PROCESS); tp_devices: one ordinal per rank
            o.tp_mode = 1
            o.tp_collective = {None: 0, "rccl": 1, "peer": 2}[tp_collective]
            if tp_devices is not None:
                if len(tp_devices) != tp_size:
                    raise ValueError("tp_devices needs tp_size entries")
                arr = (C.c_int32 * tp_size)(*[int(d) for d in tp_devices])
                o.tp_devices = C.cast(arr, C.POINTER(C.c_int32))
                keep = (keep, arr)
        return o, keep

    @classmethod
    def from_pretrained(cls, model_dir: str, **kw) -> "Model":
        """Model::new / from_pretrained (qwen3/model.rs:45-106)."""
        lib = _lib.load()
        live = cls._live_switches(kw)
        o, keep = cls._opts(**kw)
        h = C.c_void_p()
        rc = lib.cm_create(model_dir.encode(), C.byref(o), C.byref(h))
        if rc != 0:
            raise _lib.CraneError(rc, lib.cm_last_global_error().decode())
        m = cls(h, lib)
        m._apply_live(live)
        try:
            with open(f"{model_dir}/config.json") as f:
                e = json.load(f).get("eos_token_id")
            if isinstance(e, int):
                m._eos = [e]
            elif isinstance(e, list):
                m._eos = [int(x) for x in e]
        except OSError:
            pass
        return m

    @classmethod
    def synthetic(cls, config: dict, seed: int = 0, **kw) -> "Model":
        lib = _lib.load()
        live = cls._live_switches(kw)
        o, keep = cls._opts(**kw)
        h = C.c_void_p()
        rc = lib.cm_create_synthetic(json.dumps(config).encode(), seed, C.byref(o), C.byref(h))
        if rc != 0:
            raise _lib.CraneError(rc, lib.cm_last_global_error().decode())
        m = cls(h, lib)
        m._apply_live(live)
        return m

    # test / A-B switches that are not part of cm_opts: applied with cm_debug_set right after creation (before any step), so that
    # tests do not have to steer the library through the process environment
    @staticmethod
    def _live_switches(kw: dict) -> dict:
        return {k: kw.pop(k) for k in ("quant_act", "quant_prefill") if k in kw}

    def _apply_live(self, live: dict):
        if "quant_act" in live:          # "int": ggml vec_dot semantics (Q8 activations, integer dots); "f32": exact dequant x f32
            self.debug_set("quant_act_int", 0 if live["quant_act"] == "f32" else 1)
        if "quant_prefill" in live:      # False: prompts over quantised weights through the decode kernels (token-serial)
            self.debug_set("quant_prefill", 1 if live["quant_prefill"] else 0)

    def close(self):
        if self._h:
            self._lib.cm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != 0:
            raise _lib.CraneError(rc, self._lib.cm_last_error(self._h).decode())

    # -- ModelBackend (backend.rs:30-147) -------------------------------------
    def forward_step(self, input_ids: Sequence[int], start_pos: int) -> np.ndarray:
        """Logits of the last position, shape [1, 1, vocab] like qwen3."""
        arr, p = _u32(input_ids)
        out = np.empty(self.vocab_size, dtype=np.float32)
        self._check(self._lib.cm_forward_step(self._h, p, arr.size, start_pos,
                                              out.ctypes.data_as(C.POINTER(C.c_float))))
        return out.reshape(1, 1, -1)

    def forward_step_greedy(self, input_ids: Sequence[int], start_pos: int) -> int:
        arr, p = _u32(input_ids)
        t = C.c_uint32()
        self._check(self._lib.cm_forward_step_greedy(self._h, p, arr.size, start_pos, C.byref(t)))
        return int(t.value)

    def clear_kv_cache(self) -> None:
        self._lib.cm_clear_kv(self._h)

    def num_layers(self) -> int:
        return int(self._lib.cm_num_layers(self._h))

    def device(self) -> str:
        return "rocm:0"

    def dtype(self) -> str:
        return "bf16"

    def eos_token_id(self) -> List[int]:
        return list(self._eos)

    def warmup(self) -> None:
        self._check(self._lib.cm_warmup(self._h))

    def supports_kv_swap(self) -> bool:
        return True      # via paged sequences instead of tensor swaps

    def supports_batch_decode(self) -> bool:
        return True

    def active_kv_cache_bytes(self) -> int:
        return int(self._lib.cm_kv_bytes(self._h))

    # -- paged sequences (replace get/set_kv_caches + pad/stack/extract) -------
    def seq_alloc(self) -> int:
        s = C.c_int32()
        self._check(self._lib.cm_seq_alloc(self._h, C.byref(s)))
        return int(s.value)

    def seq_free(self, seq: int) -> None:
        self._check(self._lib.cm_seq_free(self._h, seq))

    def seq_fork(self, src: int) -> int:
        s = C.c_int32()
        self._check(self._lib.cm_seq_fork(self._h, src, C.byref(s)))
        return int(s.value)

    def seq_len(self, seq: int) -> int:
        return int(self._lib.cm_seq_len(self._h, seq))

    def seq_truncate(self, seq: int, new_len: int) -> None:
        self._check(self._lib.cm_seq_truncate(self._h, seq, new_len))

    def seq_forward(self, seq: int, input_ids: Sequence[int], start_pos: int, want_logits=True):
        arr, p = _u32(input_ids)
        out = np.empty(self.vocab_size, dtype=np.float32) if want_logits else None
        t = C.c_uint32()
        self._check(self._lib.cm_seq_forward(
            self._h, seq, p, arr.size, start_pos,
            out.ctypes.data_as(C.POINTER(C.c_float)) if want_logits else None, C.byref(t)))
        return out, int(t.value)

    def prefill_batch(self, seqs: Sequence[int], prompts: Sequence[Sequence[int]], want_logits=True):
        """cm_prefill_batch: whole prompts of several sequences in one pass -> ([N, V] logits of the last positions or None, greedy ids [N])."""
        n = len(seqs)
        sa = np.ascontiguousarray(np.asarray(seqs, dtype=np.int32))
        arrs = [np.ascontiguousarray(np.asarray(p, dtype=np.uint32)) for p in prompts]
        ptrs = (C.POINTER(C.c_uint32) * n)(*[a.ctypes.data_as(C.POINTER(C.c_uint32)) for a in arrs])
        lens = (C.c_size_t * n)(*[a.size for a in arrs])
        out = np.empty((n, self.vocab_size), dtype=np.float32) if want_logits else None
        g = np.empty(n, dtype=np.uint32)
        self._check(self._lib.cm_prefill_batch(
            self._h, sa.ctypes.data_as(C.POINTER(C.c_int32)), ptrs, lens, n,
            out.ctypes.data_as(C.POINTER(C.c_float)) if want_logits else None, g.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out, g

    def step_batch_decode(self, seqs: Sequence[int], tokens: Sequence[int], want_logits=True):
        """step_batch_decode (backend.rs:107-121): returns ([N,1,V] logits or None, greedy ids [N])."""
        n = len(seqs)
        sa = np.ascontiguousarray(np.asarray(seqs, dtype=np.int32))
        ta, tp_ = _u32(tokens)
        out = np.empty((n, self.vocab_size), dtype=np.float32) if want_logits else None
        g = np.empty(n, dtype=np.uint32)
        self._check(self._lib.cm_decode_batch(
            self._h, sa.ctypes.data_as(C.POINTER(C.c_int32)), tp_, n,
            out.ctypes.data_as(C.POINTER(C.c_float)) if want_logits else None,
            g.ctypes.data_as(C.POINTER(C.c_uint32))))
        return (out.reshape(n, 1, -1) if want_logits else None), g

    # -- ModelForCausalLM (based.rs:5-34; qwen3/model.rs:275-349) ---------------
    def generate(self, input_ids: Sequence[int], config: GenerationConfig,
                 streamer: Optional[TokenStreamer] = None, sync_every: int = 1) -> List[int]:
        """Returns prompt ++ generated tokens (qwen3/model.rs:348)."""
        arr, p = _u32(input_ids)
        g = _lib.CmGenConfig()
        g.max_new_tokens = config.max_new_tokens
        g.temperature = -1.0 if config.temperature is None else float(config.temperature)
        g.top_p = -1.0 if config.top_p is None else float(config.top_p)
        g.repetition_penalty = float(config.repetition_penalty)
        g.repeat_last_n = config.repeat_last_n
        eos = [config.eos_token_id] if config.eos_token_id is not None else self._eos
        for i in range(4):
            g.eos_token_id[i] = int(eos[i]) if i < len(eos) else -1
        g.sync_every = sync_every
        g.top_k = int(config.top_k)
        g.seed_lo, g.seed_hi = int(config.seed) & 0xFFFFFFFF, (int(config.seed) >> 32) & 0xFFFFFFFF
        g.frequency_penalty, g.presence_penalty = float(config.frequency_penalty), float(config.presence_penalty)
        out = np.empty(arr.size + config.max_new_tokens, dtype=np.uint32)
        n_out = C.c_size_t(0)

        def _cb(_user, tok):
            if streamer is not None:
                streamer.append(int(tok))
            return 0

        cb = _lib.TOKEN_CB(_cb)
        rc = self._lib.cm_generate(self._h, p, arr.size, C.byref(g),
                                   out.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(n_out), cb, None)
        if streamer is not None:
            streamer.finalize()
        self._check(rc)
        return [int(t) for t in out[:n_out.value]]

    # -- device sampler (crane-serve/src/engine/sampling.rs, crane_core::ops::topk_indices) --------------
    def sample(self, context: Sequence[int] = (), temperature: float = 0.0, top_p: float = 0.0, top_k: int = 0,
               repetition_penalty: float = 1.0, frequency_penalty: float = 0.0, presence_penalty: float = 0.0,
               repeat_last_n: int = 0, seed: int = 299792458, draw: int = 0) -> int:
        """Sequence::sample on the HBM-resident logits of the last forward call (penalties applied in place)."""
        sp = _lib.CmSampleParams()
        sp.temperature, sp.top_p, sp.top_k = float(temperature), float(top_p), int(top_k)
        sp.repetition_penalty, sp.frequency_penalty, sp.presence_penalty = float(repetition_penalty), float(frequency_penalty), float(presence_penalty)
        sp.repeat_last_n, sp.draw, sp.seed = int(repeat_last_n), int(draw), int(seed)
        tok = C.c_uint32(0)
        if len(context):
            arr, p = _u32(context)
            n = arr.size
        else:
            p, n = None, 0
        self._check(self._lib.cm_sample(self._h, C.byref(sp), p, n, C.byref(tok)))
        return int(tok.value)

    def topk(self, k: int, logits: Optional[np.ndarray] = None):
        """topk_indices: exact top-k (value desc, index asc).  logits None -> the last forward's logits."""
        idx = np.empty(k, dtype=np.uint32)
        val = np.empty(k, dtype=np.float32)
        if logits is not None:
            lg = np.ascontiguousarray(logits, dtype=np.float32).reshape(-1)
            lp, n = lg.ctypes.data_as(C.POINTER(C.c_float)), lg.size
        else:
            lp, n = None, 0
        self._check(self._lib.cm_topk(self._h, lp, n, k, idx.ctypes.data_as(C.POINTER(C.c_uint32)),
                                      val.ctypes.data_as(C.POINTER(C.c_float))))
        return idx, val

    def read_logits(self) -> np.ndarray:
        out = np.empty(self.vocab_size, dtype=np.float32)
        self._check(self._lib.cm_read_logits(self._h, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    # -- vision-language (qwen3_5/vision.rs, vlm.rs) ----------------------------------
    def image_token_id(self) -> int:
        return int(self._lib.cm_image_token_id(self._h))

    def encode_images(self, pixel_values: np.ndarray, grid_thw) -> np.ndarray:
        """Qwen3_5VLModel::encode_images -> [n_patches / merge^2, out_hidden] f32."""
        pv = np.ascontiguousarray(pixel_values, dtype=np.float32)
        g = np.ascontiguousarray(np.asarray(grid_thw, dtype=np.uint32).reshape(-1, 3))
        rows = C.c_size_t(0)
        out = np.empty((pv.shape[0], self.hidden_size), dtype=np.float32)       # upper bound on rows
        self._check(self._lib.cm_vision_encode(self._h, pv.ctypes.data_as(C.POINTER(C.c_float)), pv.shape[0],
                                               g.ctypes.data_as(C.POINTER(C.c_uint32)), g.shape[0],
                                               out.ctypes.data_as(C.POINTER(C.c_float)), C.byref(rows)))
        return out.reshape(-1)[: rows.value * self.hidden_size].reshape(rows.value, self.hidden_size).copy()

    def vlm_forward(self, input_ids: Sequence[int], pixel_values: np.ndarray, grid_thw, start_pos: int = 0, seq: int = 0):
        """Qwen3_5VLModel::forward (vlm.rs:250-285): logits [V] of the last position and its arg-max."""
        arr, p = _u32(input_ids)
        pv = np.ascontiguousarray(pixel_values, dtype=np.float32)
        g = np.ascontiguousarray(np.asarray(grid_thw, dtype=np.uint32).reshape(-1, 3))
        out = np.empty(self.vocab_size, dtype=np.float32)
        t = C.c_uint32()
        self._check(self._lib.cm_vlm_forward(self._h, seq, p, arr.size, start_pos, pv.ctypes.data_as(C.POINTER(C.c_float)),
                                             pv.shape[0], g.ctypes.data_as(C.POINTER(C.c_uint32)), g.shape[0],
                                             out.ctypes.data_as(C.POINTER(C.c_float)), C.byref(t)))
        return out, int(t.value)

    def embed_tokens(self, input_ids: Sequence[int]) -> np.ndarray:
        """Qwen3_5TextModel::embed_only (qwen3_5/model.rs:368-370) -> [n, hidden] f32."""
        arr, p = _u32(input_ids)
        out = np.empty((arr.size, self.hidden_size), dtype=np.float32)
        self._check(self._lib.cm_embed_tokens(self._h, p, arr.size, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def forward_embeds(self, embeds: np.ndarray, position_ids=None, start_pos: int = 0, seq: int = 0):
        """Qwen3_5TextModel::forward_embeds (qwen3_5/model.rs:430-510): logits [V] of the last row and its arg-max.
        position_ids: [3, n] (T, H, W) or None (the sequence's own counter on all axes)."""
        e = np.ascontiguousarray(embeds, dtype=np.float32).reshape(-1, self.hidden_size)
        pp = None
        if position_ids is not None:
            pos = np.ascontiguousarray(np.asarray(position_ids, dtype=np.int32).reshape(3, e.shape[0]))
            pp = pos.ctypes.data_as(C.POINTER(C.c_int32))
        out = np.empty(self.vocab_size, dtype=np.float32)
        t = C.c_uint32()
        self._check(self._lib.cm_forward_embeds(self._h, seq, e.ctypes.data_as(C.POINTER(C.c_float)), e.shape[0], pp, start_pos,
                                                out.ctypes.data_as(C.POINTER(C.c_float)), C.byref(t)))
        return out, int(t.value)

    # -- measurement hooks --------------------------------------------------------
    def bench_decode(self, first_token: int, k: int):
        toks = np.empty(k, dtype=np.uint32)
        ms = C.c_float()
        self._check(self._lib.cm_bench_decode(self._h, first_token, k,
                                              toks.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(ms)))
        return toks, float(ms.value)

    _KERNEL_NAMES = {"qkv": "gemv<rmsnorm,store>", "o": "gemv<plain,resadd>", "gate_up": "gemv<rmsnorm,silu_mul>",
                     "down": "gemv<plain,resadd>", "lm_head": "gemv<rmsnorm,argmax>",
                     "chain": "engine_kernel (persistent: whole token incl. embedding row, final norm + lm_head and arg-max partials; or one layer's o_proj+gate_up+down_proj+next QKV)"}

    def bench_kernel(self, which: str, iters: int = 360) -> dict:
        ms = C.c_float()
        b = C.c_uint64()
        self._check(self._lib.cm_bench_kernel(self._h, which.encode(), iters, C.byref(ms), C.byref(b)))
        return {"kernel": f"{which}: {self._KERNEL_NAMES.get(which, which)}", "ms": float(ms.value), "bytes": int(b.value)}

    def debug_fill_kv(self, ctx: int, seed: int = 0) -> None:
        self._check(self._lib.cm_debug_fill_kv(self._h, ctx, seed))

    def debug_set(self, key: str, value: int) -> None:
        """test hook: "no_prefill" / "quant_prefill" path switches of a live model (cm_debug_set)"""
        self._check(self._lib.cm_debug_set(self._h, key.encode(), int(value)))

    def debug_read(self, what: str, n: int) -> np.ndarray:
        out = np.empty(n, dtype=np.float32)
        self._check(self._lib.cm_debug_read(self._h, what.encode(), out.ctypes.data_as(C.POINTER(C.c_float)), n))
        return out

    def debug_qgemv(self, layer: int, which: str, x: np.ndarray, n: int) -> np.ndarray:
        xv = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty(n, dtype=np.float32)
        self._check(self._lib.cm_debug_qgemv(self._h, layer, which.encode(), xv.ctypes.data_as(C.POINTER(C.c_float)), xv.size,
                                             out.ctypes.data_as(C.POINTER(C.c_float)), n))
        return out

    def debug_qgemm(self, layer: int, which: str, x: np.ndarray, n: int) -> np.ndarray:
        """cm_debug_qgemm: rows x [rows, k] through the int8-MFMA GEMM of a quantised projection -> [rows, n]."""
        xv = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty((xv.shape[0], n), dtype=np.float32)
        self._check(self._lib.cm_debug_qgemm(self._h, layer, which.encode(), xv.ctypes.data_as(C.POINTER(C.c_float)), xv.shape[0], xv.shape[1],
                                             out.ctypes.data_as(C.POINTER(C.c_float)), n))
        return out

    def decode_bytes_per_token(self, ctx: int) -> int:
        return int(self._lib.cm_decode_bytes_per_token(self._h, ctx))

    def weight_bytes(self) -> int:
        return int(self._lib.cm_weight_bytes(self._h))

    def tp_ranks(self) -> int:
        """ranks of the RCCL communicator (1: no tensor parallelism, 0: collectives are debug no-ops)"""
        return int(self._lib.cm_tp_ranks(self._h))

    def engine_active(self) -> int:
        """0: per-projection launches; 1: persistent kernel per layer; 2: the whole token in one persistent launch"""
        return int(self._lib.cm_engine_active(self._h))


# the reference's adapter name for this family (backend.rs:609-748)
Qwen3Backend = Model
