"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- the Qwen 3.5 / 3.6 / 3.8 hybrid decoder of oracle/qwen3_5_oracle.py over
Q8_0-layout weights with ggml's quantised-activation semantics (`LinearLayer::Quantized`, crane-core/src/ops/linear.rs:18-51: the
activation row quantised to Q8_0 blocks, ggml_vec_dot_q8_0_q8_0 per output; ISQ of every linear except the Gated-Delta-Net a / b gate
projections, ops/gdn/projection.rs:78-83, through quantize_row_q8_0_ref, ops/linear.rs:83-116), TEACHER-FORCED like
oracle/qgroup_oracle.py: the device reports the activation codes every int8 projection consumed (cm_debug_set("q_capture")), this
oracle checks them against its own rounding (ties only; the token mixers' rows -- attention, chunk-parallel delta rule -- within
their kernels' error budget relative to the row, Q8TeacherForcing._quant_attn) and continues from the device's codes.
Projections that read the same rows (q / k / v, in_proj_qkv / in_proj_z, gate / up) share one record, as on the device.
A decode GROUP is one record of nb rows per projection input: row b belongs to sequence b, one oracle instance per sequence
(`RowCaptures`).  The arithmetic is oracle/c/q8_ref.c -- PARITY UNPINNED against candle / ggml themselves (absent from the image).
"""
import ctypes as C
from typing import Dict

import numpy as np

from oracle.qgroup_oracle import FMT, QMat, Q8GroupOracle, _p
from oracle.qwen3_5_oracle import F32, Qwen35Config, Qwen35Oracle
from oracle import c_oracle

QUANTISED = ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj", "in_proj_qkv", "in_proj_z", "out_proj")


def quantise_linears(weights: Dict[str, np.ndarray], isq: str = "q8_0") -> Dict[str, QMat]:
    lib = c_oracle._lib()
    lib.qc_set_threads(c_oracle.host_threads())
    return {k: QMat(lib, v, FMT[isq]) for k, v in weights.items()
            if any(k.endswith(f"{l}.weight") for l in QUANTISED) or k == "lm_head.weight"}


class RowCaptures:
    """Row b of every record of a decode group's capture list, as an iterator of one-row records."""

    def __init__(self, captures, b: int):
        self.it = iter(captures)
        self.b = b

    def __iter__(self):
        return self

    def __next__(self):
        K, codes, scales = next(self.it)
        return K, codes[self.b:self.b + 1], scales[self.b:self.b + 1]


class Q8HybridOracle(Qwen35Oracle):
    def __init__(self, cfg: dict, weights: Dict[str, np.ndarray], qmats: Dict[str, QMat], kv_dtype: str = "f32", max_pos=None,
                 stats=None):
        super().__init__(Qwen35Config.from_json(cfg), weights, kv_dtype=kv_dtype, max_pos=max_pos)
        self.qm = qmats
        self.head_captured = False       # the batched int8 head of a decode group / multi-prompt pass reports its rows; the single-row GEMV head does not
        # the checks of the dense group oracle, on a shell instance that only carries lib + stats
        self.q8 = Q8GroupOracle.__new__(Q8GroupOracle)
        self.q8.lib = c_oracle._lib()
        self.q8.stats = stats if stats is not None else dict(codes=0, flipped=0, scales=0, scale_steps=0, worst_tie=0.0)
        self.cap = None
        self.tie_tol, self.mixer_tol = 2e-3, 1e-3
        self._memo = (None, None, None)

    def _lin(self, x, name, mixer=False):
        if name not in self.qm:
            return x @ self.w[name].T
        x = np.ascontiguousarray(x, F32)
        if self._memo[0] is not x:                     # a new projection input: the next record (shared by the projections that read x)
            if self.cap is not None and mixer and self.mixer_tol is not None:
                q8, d8 = self.q8._quant_attn(x, self.cap, self.mixer_tol)
            else:
                q8, d8 = self.q8._quant(x, self.cap, self.tie_tol)
            self._memo = (x, q8, d8)
        return self.q8._mm(self._memo[1], self._memo[2], (self.qm[name],))

    def _head(self, last):
        if "lm_head.weight" not in self.qm or self.lm_head is self.embed:      # tied: the bf16 table (qwen3_5/model.rs:617-626)
            return last @ self.lm_head.T
        q8, d8 = self.q8._quant(np.ascontiguousarray(last, F32), self.cap if self.head_captured else None, self.tie_tol)
        return self.q8._mm(q8, d8, (self.qm["lm_head.weight"],))

    def forward_tf(self, ids, start_pos: int, captures):
        """forward() with the device's capture records (None: free-running on the oracle's own rounding)."""
        self.cap = iter(captures) if captures is not None else None
        self._memo = (None, None, None)
        out = self.forward(ids, start_pos)
        if self.cap is not None and not isinstance(captures, RowCaptures):
            assert next(self.cap, None) is None, "the device quantised more activation rows than the pass has projections"
        self.cap = None
        return out
