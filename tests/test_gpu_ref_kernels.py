"""The reference's OWN GPU kernels (gdn.cu, topk.cu -- compiled by oracle/build_ref.sh into oracle/_ref/*.hsaco, loaded by
oracle/ref_kernels.py) run on the MI355X beside the oracle and the product:

  * the numpy restatement of the Gated-Delta-Net recurrence (oracle/qwen3_5_oracle.gated_delta_rule, which every hybrid
    parity test of crane_amd is measured against) is pinned on the reference's fused recurrence kernel -- the reference's
    own test replayed (tests/rocm_kernels.rs:38-86: cosine >= 0.9999 on N(0,1) inputs, K = 128 and the runtime-K path at 64)
    plus a well-conditioned variant (l2-normalised q / k, as the model feeds the kernel) held to 2e-5;
  * the top-k order (value descending, index ascending, -0.0 == +0.0) of oracle/sampler_oracle.topk_indices AND of cm_topk is
    pinned on the reference's two-stage top-k kernels, bit for bit (tests/rocm_kernels.rs:88-200 inputs: vocabulary 248 320,
    k in 1 ... 512, ties, short and awkward lengths).

Skipped (not failed) where the code objects are absent: they are built from /root/reference, which a fresh clone lacks."""
import math

import numpy as np
import pytest

from crane_amd import configs
from oracle import ref_kernels
from oracle import sampler_oracle as S
from oracle.qwen3_5_oracle import gated_delta_rule, l2_norm

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not ref_kernels.available(), reason="oracle/_ref/*.hsaco not built (needs /root/reference)")]


@pytest.fixture(scope="module")
def ref():
    try:
        return ref_kernels.RefKernels()
    except Exception as e:                      # optional checker: code objects from another toolchain / no HIP runtime
        pytest.skip(f"the reference's code objects do not load here: {e}")


def _cos(a, b):
    a, b = a.reshape(-1).astype(np.float64), b.reshape(-1).astype(np.float64)
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))


def _gdn_inputs(seed, S_, H, K, V, normalise):
    r = np.random.default_rng(seed)
    q = r.standard_normal((S_, H, K)).astype(np.float32)
    k = r.standard_normal((S_, H, K)).astype(np.float32)
    if normalise:
        q, k = l2_norm(q), l2_norm(k)
    v = r.standard_normal((S_, H, V)).astype(np.float32)
    g = (0.01 * r.standard_normal((S_, H)) - 0.05).astype(np.float32)
    beta = (1.0 / (1.0 + np.exp(-r.standard_normal((S_, H))))).astype(np.float32)
    state = (0.1 * r.standard_normal((H, K, V))).astype(np.float32)
    return q, k, v, g, beta, state


def _run_both(ref, q, k, v, g, beta, state):
    K = q.shape[2]
    st = state.copy()
    y_oracle = gated_delta_rule(q, k, v, g, beta, st)                                   # [S, H, V]; st updated in place
    # kernel layout [BH, S, *], q pre-scaled by 1/sqrt(K) (rocm_kernels.rs:62-76)
    t3 = lambda a: np.ascontiguousarray(a.transpose(1, 0, 2))
    t2 = lambda a: np.ascontiguousarray(a.T)
    y_ref, st_ref = ref.gdn_recurrence(t3(q) * np.float32(1.0 / math.sqrt(K)), t3(k), t3(v), t2(g), t2(beta), state)
    return y_oracle, st, y_ref.transpose(1, 0, 2), st_ref


@pytest.mark.parametrize("K", [128, 64])
def test_gdn_oracle_against_reference_kernel_replay(ref, K):
    """rocm_kernels.rs:38-86 with the numpy restatement in the place of the portable Candle recurrence."""
    y_o, st_o, y_r, st_r = _run_both(ref, *_gdn_inputs(1, 24, 4, K, 128, normalise=False))
    assert _cos(y_r, y_o) >= 0.9999 and _cos(st_r, st_o) >= 0.9999


@pytest.mark.parametrize("K,V,S_", [(128, 128, 48), (64, 128, 17), (128, 64, 5)])
def test_gdn_oracle_against_reference_kernel_tight(ref, K, V, S_):
    y_o, st_o, y_r, st_r = _run_both(ref, *_gdn_inputs(2, S_, 6, K, V, normalise=True))
    assert np.abs(y_r - y_o).max() <= 2e-5 * max(1.0, np.abs(y_o).max())
    assert np.abs(st_r - st_o).max() <= 2e-5 * max(1.0, np.abs(st_o).max())


@pytest.fixture(scope="module")
def model():
    from crane_amd.backend import Model
    m = Model.synthetic(configs.get_config("tiny-qwen3"), seed=0, max_seq_len=64)
    yield m
    m.close()


def test_topk_three_ways_on_qwen_vocab(ref, model):
    """rocm_kernels.rs:88-131: vocabulary 248 320, N(0, 4) logits; reference kernels == host order == cm_topk."""
    v = (np.random.default_rng(3).standard_normal(248_320) * 4.0).astype(np.float32)
    for k in (1, 8, 37, 40, 63, 64, 65, 128, 512):
        want = ref.topk_indices(v, k)
        np.testing.assert_array_equal(S.topk_indices(v, k), want, err_msg=f"oracle k={k}")
        np.testing.assert_array_equal(model.topk(k, v)[0], want, err_msg=f"cm_topk k={k}")


def test_topk_ties_signed_zeros_and_awkward_lengths(ref, model):
    """rocm_kernels.rs:134-200: equal values resolve to the lowest index, -0.0 == +0.0, lengths around the block geometry."""
    r = np.random.default_rng(4)
    tied = np.zeros(10_000, np.float32); tied[[9000, 17, 4242, 3]] = 5.0
    z = np.zeros(5000, np.float32); z[::2] = -0.0
    cases = [(tied, 40), (z, 64), (np.array([0.5, -3.0, 7.25, 1.0, 7.5], np.float32), 5)]
    for n in (1, 2, 255, 256, 257, 1023, 1025, 4095, 4097, 65_537, 151_936):
        x = np.round(r.standard_normal(n) * 3.0, 1).astype(np.float32)      # coarse grid: many ties
        cases.append((x, min(n, 40)))
    for x, k in cases:
        want = ref.topk_indices(x, k)
        np.testing.assert_array_equal(S.topk_indices(x, k), want, err_msg=f"oracle n={x.size} k={k}")
        np.testing.assert_array_equal(model.topk(k, x)[0], want, err_msg=f"cm_topk n={x.size} k={k}")


def test_reference_gdn_kernel_timing_report(ref):
    """Not a parity check: the reference's fused recurrence timed on the MI355X at the shapes of the committed rocprof profiles
    of crane_amd's own GDN kernels (profiles/README.md), so the two can be read side by side.  Written to
    gpurun_out/ref_gdn_timing.json when that directory exists; the assertion only guards against a dead kernel."""
    import json, os
    rows = []
    for name, BH, S_ in [("qwen3.5-0.8b decode step", 16, 1), ("qwen3.8-27b decode step", 48, 1),
                         ("qwen3.5-0.8b 1024-token prompt", 16, 1024), ("qwen3.8-27b 1024-token prompt", 48, 1024),
                         ("bin/gdn_bench.rs shape", 16, 512)]:
        us = ref.time_gdn_recurrence(BH, S_, iters=50 if S_ == 1 else 5)
        rows.append({"case": name, "BH": BH, "S": S_, "K": 128, "V": 128, "kernel": "gdn_recurrence_f32_k128 (reference gdn.cu)",
                     "us_per_launch": round(us, 2)})
        assert 0.0 < us < 1e6
    print(json.dumps(rows, indent=1))
    if os.path.isdir("gpurun_out"):
        with open(os.path.join("gpurun_out", "ref_gdn_timing.json"), "w") as f:
            json.dump(rows, f, indent=1)
