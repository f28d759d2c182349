"""GPU: the device sampler (kernels_sample.hip) against the reference's known-answer tests
(crane-core/tests/rocm_kernels.rs:86-200) and against oracle/sampler_oracle.py -- bit-exact for indices,
penalised logits and the uniform stream's consequences (sampled token ids)."""
import numpy as np
import pytest

from crane_amd import configs
from oracle import sampler_oracle as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model():
    from crane_amd.backend import Model
    m = Model.synthetic(configs.get_config("tiny-qwen3"), seed=0, max_seq_len=256)
    yield m
    m.close()


def host_topk(v, k):
    return S.topk_indices(v, k)


def test_topk_matches_host_reference_on_qwen_vocab(model):          # rocm_kernels.rs:106-131
    n = 248_320
    i = np.arange(n, dtype=np.uint32)
    x = i * np.uint32(2_654_435_761)
    v = ((x >> np.uint32(8)).astype(np.float32) / np.float32(1 << 24)) * 40.0 - 20.0
    v = v.astype(np.float32)
    for k in [1, 20, 31, 32, 33, 37, 40, 63, 64, 128, 512]:
        idx, val = model.topk(k, v)
        np.testing.assert_array_equal(idx, host_topk(v, k), err_msg=f"k={k}")
        np.testing.assert_array_equal(val, v[idx])


def test_topk_breaks_ties_by_lowest_index(model):                   # rocm_kernels.rs:134-157
    n = 248_320
    v = (np.arange(n) % 4).astype(np.float32) * 0.5
    idx, _ = model.topk(40, v)
    np.testing.assert_array_equal(idx, np.arange(40) * 4 + 3)
    z = np.zeros(5000, np.float32); z[1::2] = -0.0
    np.testing.assert_array_equal(model.topk(64, z)[0], np.arange(64))           # -0.0 == +0.0


def test_topk_handles_short_vectors(model):                         # rocm_kernels.rs:159-171
    idx, val = model.topk(5, np.array([0.5, -3.0, 7.25, 1.0, 7.5], np.float32))
    assert idx.tolist() == [4, 2, 3, 0, 1] and val.tolist() == [7.5, 7.25, 1.0, 0.5, -3.0]


def test_topk_handles_awkward_lengths(model):                       # rocm_kernels.rs:173-200
    rng = np.random.default_rng(5)
    for n in [1, 2, 40, 1023, 1024, 1025, 4095, 4096, 4097, 12289, 65537]:
        v = rng.standard_normal(n).astype(np.float32)
        v[rng.integers(0, n, size=max(1, n // 7))] = 1.25                          # plenty of exact ties
        for k in [1, 7, 40]:
            if k > n:
                continue
            np.testing.assert_array_equal(model.topk(k, v)[0], host_topk(v, k), err_msg=f"n={n} k={k}")


def test_topk_rejects_bad_k(model):
    from crane_amd._lib import CraneError
    with pytest.raises(CraneError):
        model.topk(513, np.zeros(1000, np.float32))
    with pytest.raises(CraneError):
        model.topk(3, np.zeros(2, np.float32))


def test_penalties_in_place_bit_exact_and_greedy_pick(model):
    ids = configs.synthetic_prompt(12, model.vocab_size)
    base = model.forward_step(ids, 0)[0, 0].copy()
    np.testing.assert_array_equal(model.read_logits(), base)
    ctx = [int(np.argmax(base))] * 3 + [int(np.argmin(base)), 7, 7, 11, 10 ** 6]    # out-of-vocab id is ignored
    for kw in [dict(repetition_penalty=1.3), dict(frequency_penalty=0.4, presence_penalty=-0.2),
               dict(repetition_penalty=1.7, frequency_penalty=0.25, presence_penalty=0.1)]:
        model.forward_step(ids, 0)
        tok = model.sample(ctx, temperature=0.0, **kw)
        want = S.apply_penalties(base, [c for c in ctx if c < base.size], kw.get("repetition_penalty", 1.0),
                                 kw.get("frequency_penalty", 0.0), kw.get("presence_penalty", 0.0))
        np.testing.assert_array_equal(model.read_logits(), want)
        assert tok == int(S.topk_indices(want, 1)[0])
    # window: only the last repeat_last_n context tokens count
    model.forward_step(ids, 0)
    model.sample(ctx, temperature=0.0, repetition_penalty=2.0, repeat_last_n=2)
    np.testing.assert_array_equal(model.read_logits(), S.apply_penalties(base, [11], 2.0))


def _oracle_pick(lg, **kw):
    idx, sc = S.sample(lg, return_scores=True, **kw)
    o = np.argsort(-sc, kind="stable")
    margin = sc[o[0]] - sc[o[1]] if len(o) > 1 else np.inf
    return int(idx[o[0]]), float(margin)


def test_sampled_tokens_match_oracle_stream(model):
    ids = configs.synthetic_prompt(9, model.vocab_size)
    base = model.forward_step(ids, 0)[0, 0].copy()
    checked = 0
    for kw in [dict(temperature=0.8, top_k=40), dict(temperature=1.3, top_k=64, top_p=0.9), dict(temperature=0.7, top_p=0.5),
               dict(temperature=1.0, top_k=5, top_p=0.99), dict(temperature=1.0), dict(temperature=2.5, top_k=3)]:
        for draw in range(24):
            want, margin = _oracle_pick(base, seed=1234567891011, draw=draw, **kw)
            got = model.sample(seed=1234567891011, draw=draw, **kw)
            if margin > 1e-4:                     # logf ulp differences can only matter at a near tie
                assert got == want, (kw, draw)
                checked += 1
    assert checked > 120


def test_sampled_distribution_follows_softmax_of_topk(model):
    ids = configs.synthetic_prompt(9, model.vocab_size)
    base = model.forward_step(ids, 0)[0, 0].copy()
    T, k, n = 4.0, 8, 3000
    idx = S.topk_indices(base, k)
    p = np.exp((base[idx] - base[idx].max()) / T); p /= p.sum()
    cnt = np.zeros(k)
    pos = {int(t): i for i, t in enumerate(idx)}
    for d in range(n):
        cnt[pos[model.sample(temperature=T, top_k=k, seed=42, draw=d)]] += 1
    chi2 = float((((cnt - n * p) ** 2) / (n * p)).sum())
    assert chi2 < 30.0, (chi2, cnt, n * p)        # 7 dof: P(chi2 > 30) ~ 1e-4 (u is truncated to [1e-7, 0.999) like the reference)


def test_generate_with_sampling(model):
    from crane_amd.backend import GenerationConfig
    ids = configs.synthetic_prompt(10, model.vocab_size)
    g = GenerationConfig(max_new_tokens=24, temperature=30.0, top_p=0.95, top_k=40, seed=7)
    a = model.generate(ids, g)
    assert a[:10] == list(ids) and len(a) == 34 and all(0 <= t < model.vocab_size for t in a)
    assert model.generate(ids, g) == a                                           # same seed -> same tokens
    g2 = GenerationConfig(max_new_tokens=24, temperature=30.0, top_p=0.95, top_k=40, seed=8)
    assert model.generate(ids, g2) != a
    assert a != model.generate(ids, GenerationConfig.greedy(24))
    greedy = model.generate(ids, GenerationConfig.greedy(24))
    assert model.generate(ids, GenerationConfig(max_new_tokens=24, temperature=1e-9, top_p=None)) == greedy
    assert model.generate(ids, GenerationConfig(max_new_tokens=24, temperature=0.5, top_k=1, top_p=None)) == greedy


def test_generate_repetition_penalty_matches_host_loop(model):
    """qwen3/model.rs:306-315: apply_repeat_penalty (true division) over the last repeat_last_n tokens, then arg-max."""
    from crane_amd.backend import GenerationConfig
    ids = list(configs.synthetic_prompt(10, model.vocab_size))
    g = GenerationConfig(max_new_tokens=12, temperature=None, top_p=None, repetition_penalty=1.5, repeat_last_n=6)
    got = model.generate(ids, g)
    toks = list(ids)
    lg = model.forward_step(toks, 0)[0, 0]
    for _ in range(12):
        pen = S.apply_penalties(lg, toks[-6:], 1.5, true_div=True)
        toks.append(int(S.topk_indices(pen, 1)[0]))
        lg = model.forward_step([toks[-1]], len(toks) - 1)[0, 0]
    assert got == toks
