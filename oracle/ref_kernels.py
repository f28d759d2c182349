"""TEST INFRASTRUCTURE ONLY -- the reference's own GPU kernels, run on the MI355X beside ours.

`oracle/build_ref.sh` compiles `crane-core/kernels/cuda/gdn.cu` and `topk.cu` of the reference (from where they lie under
/root/reference; nothing is copied) into `oracle/_ref/{gdn,topk}.hsaco`.  This module loads those code objects through the HIP
module API (ctypes on libamdhip64) and launches them with the geometry of the reference's ROCm launchers:

  * `gdn_recurrence(q, k, v, g, beta, state)`  -- `gdn_recurrence_rocm` (ops/gdn/rocm_backend.rs:34-133): grid = BH blocks of
    V threads (V_TILE = V), 2*K*4 bytes of dynamic LDS, kernel `gdn_recurrence_f32_k128` for K = 128 and the runtime-K
    `gdn_recurrence_f32` otherwise; q arrives pre-scaled by 1/sqrt(K) (gdn.cu:23-24).
  * `topk_indices(x, k)`                       -- `topk_indices` (ops/fused_ops/rocm_impl.rs:56-59,77-150): stage 1 on
    grid = clamp(ceil(n / 1024), 1, 256) blocks of 256 threads over ceil(n / grid) items each, stage 2 on one block.

Used by tests/test_gpu_ref_kernels.py to pin (a) the numpy restatement of the recurrence (`qwen3_5_oracle.gated_delta_rule`),
(b) the top-k order of `sampler_oracle.topk_indices` and (c) `cm_topk` of the product on the real reference code.  Only tests
may import this module; the product never does.
"""
import ctypes as C
import math
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(_HERE, "_ref")


def available() -> bool:
    return all(os.path.exists(os.path.join(REF_DIR, f + ".hsaco")) for f in ("gdn", "topk"))


class HipError(RuntimeError):
    pass


class _Hip:
    def __init__(self):
        self.lib = C.CDLL("libamdhip64.so")
        L = self.lib
        L.hipGetErrorString.restype = C.c_char_p
        L.hipGetErrorString.argtypes = [C.c_int]
        L.hipModuleLoad.argtypes = [C.POINTER(C.c_void_p), C.c_char_p]
        L.hipModuleGetFunction.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_char_p]
        L.hipModuleLaunchKernel.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint,
                                            C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        L.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        L.hipFree.argtypes = [C.c_void_p]
        L.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        L.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        L.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
        L.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
        L.hipEventSynchronize.argtypes = [C.c_void_p]
        L.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
        L.hipEventDestroy.argtypes = [C.c_void_p]
        self.ck(L.hipSetDevice(0))

    def ck(self, rc):
        if rc != 0:
            raise HipError(f"HIP error {rc}: {self.lib.hipGetErrorString(rc).decode()}")

    def module(self, path):
        m = C.c_void_p()
        self.ck(self.lib.hipModuleLoad(C.byref(m), path.encode()))
        return m

    def function(self, module, name):
        f = C.c_void_p()
        self.ck(self.lib.hipModuleGetFunction(C.byref(f), module, name.encode()))
        return f

    def to_device(self, a: np.ndarray) -> C.c_void_p:
        a = np.ascontiguousarray(a)
        p = C.c_void_p()
        self.ck(self.lib.hipMalloc(C.byref(p), max(a.nbytes, 4)))
        if a.nbytes:
            self.ck(self.lib.hipMemcpy(p, a.ctypes.data_as(C.c_void_p), a.nbytes, 1))        # hipMemcpyHostToDevice
        return p

    def alloc(self, nbytes: int) -> C.c_void_p:
        p = C.c_void_p()
        self.ck(self.lib.hipMalloc(C.byref(p), max(nbytes, 4)))
        self.ck(self.lib.hipMemset(p, 0, max(nbytes, 4)))
        return p

    def to_host(self, p, shape, dtype) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        self.ck(self.lib.hipDeviceSynchronize())
        if out.nbytes:
            self.ck(self.lib.hipMemcpy(out.ctypes.data_as(C.c_void_p), p, out.nbytes, 2))    # hipMemcpyDeviceToHost
        return out

    def free(self, *ptrs):
        for p in ptrs:
            self.lib.hipFree(p)

    def launch(self, fn, grid, block, shared, args, sync=True):
        """args: ctypes scalars / c_void_p device pointers, passed by address like the reference's `arg(&x)` list."""
        arr = (C.c_void_p * len(args))(*[C.cast(C.pointer(a), C.c_void_p) for a in args])
        gx, gy = (grid if isinstance(grid, tuple) else (grid, 1))
        self.ck(self.lib.hipModuleLaunchKernel(fn, gx, gy, 1, block, 1, 1, shared, None, arr, None))
        if sync:
            self.ck(self.lib.hipDeviceSynchronize())

    def time_launches(self, fn, grid, block, shared, args, iters, warmup=3) -> float:
        """Average microseconds per launch of `iters` back-to-back launches on the null stream (HIP events)."""
        L = self.lib
        e0, e1 = C.c_void_p(), C.c_void_p()
        self.ck(L.hipEventCreate(C.byref(e0))); self.ck(L.hipEventCreate(C.byref(e1)))
        for _ in range(warmup):
            self.launch(fn, grid, block, shared, args, sync=False)
        self.ck(L.hipDeviceSynchronize())
        self.ck(L.hipEventRecord(e0, None))
        for _ in range(iters):
            self.launch(fn, grid, block, shared, args, sync=False)
        self.ck(L.hipEventRecord(e1, None))
        self.ck(L.hipEventSynchronize(e1))
        ms = C.c_float(0)
        self.ck(L.hipEventElapsedTime(C.byref(ms), e0, e1))
        L.hipEventDestroy(e0); L.hipEventDestroy(e1)
        return float(ms.value) * 1e3 / iters


class RefKernels:
    """The reference's gdn.cu / topk.cu kernels on cuda:0."""

    def __init__(self):
        if not available():
            raise FileNotFoundError("oracle/_ref/*.hsaco missing: run oracle/build_ref.sh where /root/reference exists")
        self.hip = _Hip()
        self.m_gdn = self.hip.module(os.path.join(REF_DIR, "gdn.hsaco"))
        self.m_topk = self.hip.module(os.path.join(REF_DIR, "topk.hsaco"))
        self.f_gdn_k128 = self.hip.function(self.m_gdn, "gdn_recurrence_f32_k128")
        self.f_gdn = self.hip.function(self.m_gdn, "gdn_recurrence_f32")
        self.f_topk1 = self.hip.function(self.m_topk, "topk_stage1_f32")
        self.f_topk2 = self.hip.function(self.m_topk, "topk_stage2_u64")

    def gdn_recurrence(self, q, k, v, g, beta, state):
        """q, k [BH, S, K] (q already scaled by 1/sqrt(K)), v [BH, S, V], g, beta [BH, S], state [BH, K, V], all f32.
        Returns (y [BH, S, V], state_out [BH, K, V])."""
        q, k, v, g, beta, state = (np.ascontiguousarray(a, dtype=np.float32) for a in (q, k, v, g, beta, state))
        BH, S, K = q.shape
        V = v.shape[2]
        assert K <= 256 and k.shape == (BH, S, K) and v.shape == (BH, S, V) and g.shape == (BH, S) and beta.shape == (BH, S)
        assert state.shape == (BH, K, V)
        h = self.hip
        dq, dk, dv, dg, db, ds = (h.to_device(a) for a in (q, k, v, g, beta, state))
        dy, dso = h.alloc(BH * S * V * 4), h.alloc(BH * K * V * 4)
        args = [dq, dk, dv, dg, db, ds, dso, dy, C.c_int(BH), C.c_int(S)]
        fn = self.f_gdn_k128
        if K != 128:
            args.append(C.c_int(K))
            fn = self.f_gdn
        args += [C.c_int(V), C.c_int(V)]                                  # V, V_TILE = V (one block per head)
        try:
            h.launch(fn, BH, V, 2 * K * 4, args)
            return h.to_host(dy, (BH, S, V), np.float32), h.to_host(dso, (BH, K, V), np.float32)
        finally:
            h.free(dq, dk, dv, dg, db, ds, dy, dso)

    def time_gdn_recurrence(self, BH, S, K=128, V=128, iters=20) -> float:
        """Microseconds per launch of the reference recurrence on bin/gdn_bench.rs-style inputs (state in != state out)."""
        r = np.random.default_rng(0)
        q = (r.standard_normal((BH, S, K)) / math.sqrt(K)).astype(np.float32)
        k = r.standard_normal((BH, S, K)).astype(np.float32)
        k /= np.linalg.norm(k, axis=-1, keepdims=True)
        v = r.standard_normal((BH, S, V)).astype(np.float32)
        g = (0.01 * r.standard_normal((BH, S)) - 0.05).astype(np.float32)
        beta = (1.0 / (1.0 + np.exp(-r.standard_normal((BH, S))))).astype(np.float32)
        h = self.hip
        dq, dk, dv, dg, db = (h.to_device(a) for a in (q, k, v, g, beta))
        ds, dso, dy = h.alloc(BH * K * V * 4), h.alloc(BH * K * V * 4), h.alloc(BH * S * V * 4)
        args = [dq, dk, dv, dg, db, ds, dso, dy, C.c_int(BH), C.c_int(S)]
        fn = self.f_gdn_k128
        if K != 128:
            args.append(C.c_int(K)); fn = self.f_gdn
        args += [C.c_int(V), C.c_int(V)]
        try:
            return h.time_launches(fn, BH, V, 2 * K * 4, args, iters)
        finally:
            h.free(dq, dk, dv, dg, db, ds, dso, dy)

    def topk_indices(self, x, k):
        """Indices of the k largest values of the 1-D f32 vector x: value descending, index ascending."""
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1)
        n = x.size
        assert 0 < k <= min(n, 2048 // 4)
        grid = min(max(math.ceil(n / (256 * 4 * 4)), 1), 256)             # topk_geometry: n.div_ceil(TOPK_STEP * 4).clamp(1, 256)
        items = math.ceil(n / grid)
        h = self.hip
        dx, dkeys, dout = h.to_device(x), h.alloc(grid * k * 8), h.alloc(k * 4)
        try:
            h.launch(self.f_topk1, grid, 256, 0, [dx, C.c_uint32(n), C.c_uint32(k), C.c_uint32(items), dkeys])
            h.launch(self.f_topk2, 1, 256, 0, [dkeys, C.c_uint32(grid * k), C.c_uint32(k), dout])
            return h.to_host(dout, (k,), np.uint32)
        finally:
            h.free(dx, dkeys, dout)
