"""CPU: pin oracle/sampler_oracle.py against the reference's own known-answer tests
(crane-serve/src/engine/sampling.rs:493-641 penalties, crane-core/tests/rocm_kernels.rs:86-200 top-k)."""
import numpy as np
import pytest

from oracle import sampler_oracle as S


def P(v, ctx, rp=1.0, fp=0.0, pp=0.0):
    return S.apply_penalties(np.array(v, np.float32), ctx, rp, fp, pp)


def test_penalty_kats_from_reference():
    f = np.float32
    assert P([1, 2, 3], [0, 1, 2]).tolist() == [1, 2, 3]                       # all_penalties_noop_when_disabled
    assert P([1, 2, 3], [], 1.1, 0.5, 0.5).tolist() == [1, 2, 3]               # ..._when_context_empty
    np.testing.assert_allclose(P([10, -10, 3], [0, 1], 2.0), [5, -20, 3], atol=1e-6)          # scales_by_sign
    np.testing.assert_allclose(P([10, 10], [0, 0, 1], 2.0, 1.0), [3, 4], atol=1e-6)           # combine_in_order
    assert P([10, 10, 10], [0, 0, 0, 1], 1.0, 0.5).tolist() == [f(10) - f(1.5), f(10) - f(0.5), 10]   # scales_with_count
    assert P([10, 10, 10], [0, 0, 0, 1], 1.0, 0.0, 0.5).tolist() == [9.5, 9.5, 10]             # presence flat
    out = P([5.0, 4.9], [0, 1, 0, 0, 0, 0], 1.0, 0.1)                                          # breaks_short_repeat_cycle
    assert abs(out[0] - 4.5) < 1e-6 and abs(out[1] - 4.8) < 1e-6 and out[1] > out[0]
    assert P([10, 10, 10], [0, 0, 0, 1], 1.0, 0.5, 0.2).tolist() == [f(10) - (f(3) * f(0.5) + f(0.2)), f(10) - (f(0.5) + f(0.2)), 10]
    np.testing.assert_allclose(P([10, 10], [0, 0, 1], 1.0, -0.5), [11, 10.5], atol=1e-6)       # negative frequency
    np.testing.assert_allclose(P([10, 10, 10], [0, 1], 1.0, 0.0, -0.5), [10.5, 10.5, 10], atol=1e-6)


def test_penalty_true_division_variant_and_out_of_range_ids():
    v = np.array([3.0, -1.0, 7.0], np.float32)
    a = S.apply_penalties(v, [0, 2, 99], 1.3, true_div=True)
    assert a[0] == np.float32(3.0) / np.float32(1.3) and a[1] == -1.0 and a[2] == np.float32(7.0) / np.float32(1.3)
    b = S.apply_penalties(v, [0], 1.3)
    assert b[0] == np.float32(3.0) * np.float32(1.0 / 1.3)


def test_topk_kats_from_reference():
    # topk_breaks_ties_by_lowest_index (rocm_kernels.rs:134-157)
    n = 4096
    v = (np.arange(n) % 4).astype(np.float32) * 0.5
    np.testing.assert_array_equal(S.topk_indices(v, 40), np.arange(40) * 4 + 3)
    # topk_handles_short_vectors (rocm_kernels.rs:159-171)
    np.testing.assert_array_equal(S.topk_indices([0.5, -3.0, 7.25, 1.0, 7.5], 5), [4, 2, 3, 0, 1])
    # -0.0 and +0.0 tie -> lower index first
    np.testing.assert_array_equal(S.topk_indices([0.0, -0.0, -0.0, 0.0], 4), [0, 1, 2, 3])


def test_uniform_stream_is_deterministic_and_in_range():
    u = S.uniform_stream(299792458, 3, 1 << 16)
    assert u.dtype == np.float32 and u.min() >= np.float32(1e-7) and u.max() < np.float32(0.999)
    np.testing.assert_array_equal(u, S.uniform_stream(299792458, 3, 1 << 16))
    assert not np.array_equal(u[:64], S.uniform_stream(299792458, 4, 64))
    assert not np.array_equal(u[:64], S.uniform_stream(299792459, 3, 64))
    assert abs(float(u.mean()) - 0.4995) < 5e-3                     # U(1e-7, 0.999)


def test_topp_mask_rule():
    # probs 0.5, 0.3, 0.15, 0.05: keep i if cumsum[i] <= p or cumsum[i-1] <= p (sampling.rs:311-322)
    lg = np.log(np.array([0.5, 0.3, 0.15, 0.05], np.float32))
    assert S.topp_mask(lg, 1.0, 0.81).tolist() == [True, True, True, False]
    assert S.topp_mask(lg, 1.0, 0.6).tolist() == [True, True, False, False]
    assert S.topp_mask(lg, 1.0, 0.3).tolist() == [True, False, False, False]      # degenerate: best token kept


def test_sample_rules():
    rng = np.random.default_rng(0)
    lg = rng.standard_normal(1000).astype(np.float32)
    assert S.sample(lg, temperature=0.0) == int(np.argmax(lg))
    assert S.sample(lg, temperature=0.7, top_k=1) == int(np.argmax(lg))
    top64 = set(S.topk_indices(lg, 64).tolist())
    picks = {S.sample(lg, temperature=1.5, top_p=0.95, draw=d) for d in range(200)}
    assert picks <= top64 and len(picks) > 5                                        # top_k==0 & top_p -> 64 (sampling.rs:263)
    # Gumbel-max draws follow softmax(logits / T)
    small = np.array([2.0, 1.0, 0.0, -1.0], np.float32)
    cnt = np.bincount([S.sample(small, temperature=1.0, draw=d) for d in range(4000)], minlength=4)
    p = np.exp(small) / np.exp(small).sum()
    assert np.abs(cnt / 4000 - p).max() < 0.03
