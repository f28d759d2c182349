"""ctypes wrapper of oracle/c/libqwen3_cpu.so (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).

``time_decode`` is bench.py's ``cpu_baseline`` leg: the C port of the reference's CPU decode
path (kind "port": the reference itself needs a Rust toolchain + candle 0.11, both absent) timed
on the host cores for a bounded number of decode steps of the SAME workload.
"""
from __future__ import annotations

import ctypes as C
import os
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(_HERE, "c", "libqwen3_cpu.so")


class QcCfg(C.Structure):
    _fields_ = [("V", C.c_int), ("H", C.c_int), ("I", C.c_int), ("L", C.c_int), ("Hq", C.c_int),
                ("Hkv", C.c_int), ("D", C.c_int), ("max_seq", C.c_int), ("eps", C.c_float),
                ("theta", C.c_double), ("tie", C.c_int), ("qk_norm", C.c_int), ("kv_bf16", C.c_int)]   # kv_bf16: 0 f32, 1 bf16, 2 f16 K/V rounding


class Q5Cfg(C.Structure):
    _fields_ = [("V", C.c_int), ("H", C.c_int), ("I", C.c_int), ("L", C.c_int), ("Hq", C.c_int),
                ("Hkv", C.c_int), ("D", C.c_int), ("max_seq", C.c_int), ("eps", C.c_float),
                ("theta", C.c_double), ("tie", C.c_int), ("rot_dim", C.c_int), ("interval", C.c_int), ("NK", C.c_int),
                ("NV", C.c_int), ("Kd", C.c_int), ("Vd", C.c_int), ("conv_k", C.c_int), ("kv_round", C.c_int)]


def _lib():
    lib = C.CDLL(SO)
    lib.q5_create.argtypes = [C.POINTER(Q5Cfg), C.c_uint64]
    lib.q5_create.restype = C.c_void_p
    lib.q5_destroy.argtypes = [C.c_void_p]
    lib.q5_forward.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_int, C.c_int, C.POINTER(C.c_float)]
    lib.q5_fill_kv_paged.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_int]
    lib.qc_create.argtypes = [C.POINTER(QcCfg), C.c_uint64]
    lib.qc_create.restype = C.c_void_p
    lib.qc_destroy.argtypes = [C.c_void_p]
    lib.qc_forward.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_int, C.c_int, C.POINTER(C.c_float)]
    lib.qc_forward_batched.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_int, C.c_int, C.POINTER(C.c_float)]
    lib.qc_fill_kv.argtypes = [C.c_void_p, C.c_int, C.c_uint64]
    lib.qc_fill_kv_paged.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_int]
    lib.qc_num_threads.restype = C.c_int
    lib.qc_set_threads.argtypes = [C.c_int]
    lib.qc_set_name_prefix.argtypes = [C.c_char_p]
    if hasattr(lib, "qc_synth_f32"):          # oracle/c/q8_ref.c
        lib.qc_synth_f32.argtypes = [C.c_char_p, C.c_uint64, C.c_double, C.c_float, C.c_int, C.c_int, C.POINTER(C.c_float)]
        lib.qc_quantize_ref.argtypes = [C.c_int, C.POINTER(C.c_float), C.c_size_t, C.POINTER(C.c_int8), C.POINTER(C.c_float)]
        lib.qc_vec_dot_q8_rows.argtypes = [C.POINTER(C.c_int8), C.POINTER(C.c_float), C.c_int, C.c_int, C.POINTER(C.c_int8),
                                           C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float)]
    return lib


def host_threads() -> int:
    """OpenMP team size for the C port: the CPUs this process may actually use -- min(affinity mask, cgroup CPU quota).
    The GPU hosts show 256 logical CPUs but run the container under a 16-CPU quota; 128 threads there were CFS-throttled
    into 1-2 tokens/s with 2x run-to-run noise, 16 threads stream ~350 GB/s (tools/cpu_probe.py).  QC_THREADS overrides."""
    env = os.environ.get("QC_THREADS")
    if env:
        return max(1, int(env))
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p_))
        except (OSError, ValueError):
            pass
    return max(1, min(n, 64))


class CQwen3:
    def __init__(self, cfg: dict, seed: int = 0, max_seq: int = 2048, kv_bf16: bool = False):
        self.lib = _lib()
        self.lib.qc_set_threads(host_threads())
        if "text_config" in cfg:         # the dense text model of a vision-language checkpoint (Qwen3-VL): tensors under model.language_model.
            tie = cfg.get("tie_word_embeddings", True)
            rp = cfg["text_config"].get("rope_parameters") or {}
            cfg = dict(cfg["text_config"], tie_word_embeddings=cfg["text_config"].get("tie_word_embeddings", tie))
            cfg.setdefault("rope_theta", rp.get("rope_theta", 1e6))
            self.lib.qc_set_name_prefix(b"model.language_model.")
        else:
            self.lib.qc_set_name_prefix(b"model.")
        D = cfg.get("head_dim") or cfg["hidden_size"] // cfg["num_attention_heads"]
        c = QcCfg(cfg["vocab_size"], cfg["hidden_size"], cfg["intermediate_size"], cfg["num_hidden_layers"],
                  cfg["num_attention_heads"], cfg["num_key_value_heads"], D, max_seq,
                  cfg.get("rms_norm_eps", 1e-6), cfg.get("rope_theta", 1e6),
                  int(cfg.get("tie_word_embeddings", True)), int(cfg.get("use_qk_norm", True)), int(kv_bf16))
        self.V = cfg["vocab_size"]
        self.h = self.lib.qc_create(C.byref(c), seed)

    def forward(self, ids, start_pos: int) -> np.ndarray:
        a = np.ascontiguousarray(np.asarray(ids, dtype=np.uint32))
        out = np.empty(self.V, dtype=np.float32)
        rc = self.lib.qc_forward(self.h, a.ctypes.data_as(C.POINTER(C.c_uint32)), a.size, start_pos,
                                 out.ctypes.data_as(C.POINTER(C.c_float)))
        if rc != 0:
            raise RuntimeError(f"qc_forward failed: {rc}")
        return out

    def forward_batched(self, ids, start_pos: int) -> np.ndarray:
        """qc_forward with every weight matrix streamed once for the whole prompt (layer-major); bit-identical to forward()."""
        a = np.ascontiguousarray(np.asarray(ids, dtype=np.uint32))
        out = np.empty(self.V, dtype=np.float32)
        rc = self.lib.qc_forward_batched(self.h, a.ctypes.data_as(C.POINTER(C.c_uint32)), a.size, start_pos,
                                         out.ctypes.data_as(C.POINTER(C.c_float)))
        if rc != 0:
            raise RuntimeError(f"qc_forward_batched failed: {rc}")
        return out

    def fill_kv(self, ctx: int, seed: int = 1):
        self.lib.qc_fill_kv(self.h, ctx, seed)

    def fill_kv_paged(self, ctx: int, seed: int = 1, page: int = 64):
        """Same values as cm_debug_fill_kv (the device fills its pool page by page)."""
        self.lib.qc_fill_kv_paged(self.h, ctx, seed, page)

    def threads(self) -> int:
        return int(self.lib.qc_num_threads())

    def close(self):
        if self.h:
            self.lib.qc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CQwen35(CQwen3):
    """oracle/c/qwen35_cpu.c: the Qwen 3.5 / 3.6 / 3.8 hybrid decode path (Gated Delta Net + gated attention), token-serial."""

    def __init__(self, cfg: dict, seed: int = 0, max_seq: int = 2048, kv_round: int = 0):
        self.lib = _lib()
        self.lib.qc_set_threads(host_threads())
        t = cfg.get("text_config", cfg)
        rp = t.get("rope_parameters") or {}
        D = t["head_dim"]
        tie = t.get("tie_word_embeddings", cfg.get("tie_word_embeddings", False))
        c = Q5Cfg(t["vocab_size"], t["hidden_size"], t["intermediate_size"], t["num_hidden_layers"], t["num_attention_heads"],
                  t["num_key_value_heads"], D, max_seq, t.get("rms_norm_eps", 1e-6), rp.get("rope_theta", t.get("rope_theta", 1e7)),
                  int(tie), int(D * rp.get("partial_rotary_factor", t.get("partial_rotary_factor", 0.25))),
                  t.get("full_attention_interval", 4), t["linear_num_key_heads"], t["linear_num_value_heads"],
                  t["linear_key_head_dim"], t["linear_value_head_dim"], t.get("linear_conv_kernel_dim", 4), int(kv_round))
        self.V = t["vocab_size"]
        self.h = self.lib.q5_create(C.byref(c), seed)

    def forward(self, ids, start_pos: int) -> np.ndarray:
        a = np.ascontiguousarray(np.asarray(ids, dtype=np.uint32))
        out = np.empty(self.V, dtype=np.float32)
        rc = self.lib.q5_forward(self.h, a.ctypes.data_as(C.POINTER(C.c_uint32)), a.size, start_pos,
                                 out.ctypes.data_as(C.POINTER(C.c_float)))
        if rc != 0:
            raise RuntimeError(f"q5_forward failed: {rc}")
        return out

    def fill_kv(self, ctx: int, seed: int = 1):
        raise NotImplementedError

    def fill_kv_paged(self, ctx: int, seed: int = 1, page: int = 64):
        self.lib.q5_fill_kv_paged(self.h, ctx, seed, page)

    def close(self):
        if self.h:
            self.lib.q5_destroy(self.h)
            self.h = None


def time_decode(model_name: str, ctx: int, budget_s: float = 20.0, prompt_len: int = 48, n_new: int = 16, long_ctx: int = 0,
                long_budget_s: float = 90.0):
    """bench.py's CPU leg: (cpu_baseline dict, greedy tokens, logits of the first step, model-written-cache reference).

    Timing: the KV cache of positions [0, ctx) holds the values cm_debug_fill_kv writes on the device and the first token is
    bench.py's (3) -- the SAME workload the GPU is timed on; the tokens / first logits of that run are returned too, but they
    only check the kernels at the timed shapes: the filled values are bf16-exact on both sides, so they say nothing about
    what the device's K/V pages do to values the model wrote itself.
    Parity proper (4th result): a `prompt_len`-token prompt fed from an EMPTY cache, one decode step and `n_new` greedy tokens
    -- the f32 CPU forward (K/V appends unrounded) on a cache the model wrote itself; bench.py runs the same through the
    HIP path (MFMA prefill, then the decode kernels over the pages that prefill wrote) and compares logits and ids.
    `long_ctx` > 0 adds written["long"]: the same on a `long_ctx`-token prompt (the benchmark's context: K/V rounding error of the
    device's pages grows with depth AND context, so the headline configuration is checked at its own context, not extrapolated
    from 48 tokens).  The dense port streams each weight matrix once for the whole prompt (qc_forward_batched, bit-identical to
    the token-serial forward); the hybrid port is token-serial and the leg is skipped -- with the reason -- when the measured
    decode rate says it would take longer than `long_budget_s`."""
    from crane_amd import configs
    cfg = configs.get_config(model_name)
    t0 = time.perf_counter()
    hybrid = cfg.get("text_config", cfg).get("model_type", cfg.get("model_type", "qwen3")).startswith("qwen3_5")
    m = CQwen35(cfg, seed=0, max_seq=ctx + 64) if hybrid else CQwen3(cfg, seed=0, max_seq=ctx + 64, kv_bf16=False)
    t_build = time.perf_counter() - t0
    prompt = configs.synthetic_prompt(prompt_len, cfg.get("text_config", cfg)["vocab_size"])
    p_logits = m.forward(prompt, 0)
    gen = [int(p_logits.argmax())]
    d_logits = m.forward([gen[0]], prompt_len)                       # the decode step right after the prompt
    lg = d_logits
    for i in range(1, n_new):
        gen.append(int(lg.argmax()))
        lg = m.forward([gen[-1]], prompt_len + i)
    written = {"prompt": prompt, "prefill_logits": p_logits, "decode_logits": d_logits, "greedy": gen}
    m.fill_kv_paged(ctx, 1, 64)
    first = m.forward([3], ctx)                # untimed first step (page-in); its logits are the parity reference
    tok = int(first.argmax())
    toks = [tok]
    n, dt = 0, 0.0
    while n < 32 and dt < budget_s:
        t1 = time.perf_counter()
        lg = m.forward([tok], ctx + 1 + n)
        dt += time.perf_counter() - t1
        tok = int(lg.argmax())
        toks.append(tok)
        n += 1
    thr = m.threads()
    if long_ctx > 0:
        est = long_ctx / max(n / dt, 1e-9) if hybrid else 0.0
        if hybrid and est > long_budget_s:
            written["long"] = {"skipped": f"token-serial hybrid CPU port: {long_ctx} tokens at {n / dt:.1f} tokens/s = {est:.0f}s "
                                          f"> {long_budget_s:.0f}s budget (run bench.py --parity-budget {int(est) + 60} to include it)"}
        else:
            t1 = time.perf_counter()
            lp = configs.synthetic_prompt(long_ctx, cfg.get("text_config", cfg)["vocab_size"])
            pl = m.forward(lp, 0) if hybrid else m.forward_batched(lp, 0)
            g2 = [int(pl.argmax())]
            dl = m.forward([g2[0]], long_ctx)
            lg2 = dl
            for i in range(1, n_new):
                g2.append(int(lg2.argmax()))
                lg2 = m.forward([g2[-1]], long_ctx + i)
            written["long"] = {"prompt": lp, "prefill_logits": pl, "decode_logits": dl, "greedy": g2,
                               "cpu_seconds": round(time.perf_counter() - t1, 1)}
    m.close()
    base = {"value": round(n / dt, 3), "unit": "tokens/s", "cores": thr, "kind": "port",
            "sample": f"{n} greedy decode steps of {model_name} at context {ctx} (bf16-stored weights, f32 compute, "
                      f"OpenMP over {thr} host threads; weight synthesis {t_build:.1f}s excluded)"}
    return base, toks, first, written
