"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference sampler (never imported by crane_amd/).

Follows crane-serve/src/engine/sampling.rs:
  apply_penalties      :422-478  (distinct-token counts; multiplicative repetition penalty first -- candle's
                                  `Tensor / f64` is a multiply by (f32)(1/rp) -- then count*freq + presence subtracted)
  topk_indices         crane-core/src/ops (CPU fallback = stable sort, value descending / index ascending;
                       pinned by crane-core/tests/rocm_kernels.rs:86-200 and, bit for bit, on the reference's own
                       top-k kernels built into oracle/_ref/ -- tests/test_gpu_ref_kernels.py)
  sample               :169-373  (greedy at temperature <= 0; top_k==0 with top_p -> 64; top_k = min(top_k, 64, vocab);
                                  softmax(topk/T), cumsum, keep i if cumsum[i] <= p or cumsum[i-1] <= p; Gumbel-max)
  sample_gumbel_max_idx:382-392  (u ~ U(1e-7, 0.999), argmax(logits/T - log(-log u)))

PARITY UNPINNED for the random stream: the reference draws u from candle's device RNG and pins no sampled token in
any test.  This oracle pins crane_amd's own counter-based stream (fmix32 hash of seed, draw, lane) so that the GPU
sampler is checked bit-for-bit on the uniforms and statistically on the draws.
"""
import numpy as np

M32 = 0xFFFFFFFF


def fmix32(h):
    h = np.asarray(h, dtype=np.uint64) & M32
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & M32
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & M32
    h ^= h >> 16
    return h


def uniform_stream(seed: int, draw: int, n: int) -> np.ndarray:
    """kernels_sample.hip uniform_open(): u in [1e-7, 0.999)."""
    i = np.arange(n, dtype=np.uint64)
    lo, hi = seed & M32, (seed >> 32) & M32
    h = fmix32(np.uint64(lo) ^ ((i * 0x9E3779B1) & M32))
    h = fmix32(h ^ np.uint64(hi) ^ np.uint64((draw * 0x85EBCA6B) & M32))
    u01 = (h >> 8).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return (np.float32(1e-7) + u01 * np.float32(np.float32(0.999) - np.float32(1e-7))).astype(np.float32)


def apply_penalties(logits, context, repetition_penalty=1.0, frequency_penalty=0.0, presence_penalty=0.0, true_div=False):
    """sampling.rs:422-478.  true_div=True is candle_transformers::utils::apply_repeat_penalty (model.rs:306-315)."""
    out = np.array(logits, dtype=np.float32, copy=True)
    rp, fp, pp = np.float32(repetition_penalty), np.float32(frequency_penalty), np.float32(presence_penalty)
    rep = rp != np.float32(1.0)
    fpa = fp != 0 or pp != 0
    if len(context) == 0 or not (rep or fpa):
        return out
    ids, counts = np.unique(np.asarray(context, dtype=np.int64), return_counts=True)
    keep = ids < out.size
    ids, counts = ids[keep], counts[keep]
    sel = out[ids]
    if rep:
        inv = np.float32(1.0 / float(rp))
        pos = (sel / rp) if true_div else (sel * inv)
        sel = np.where(sel >= 0, pos, sel * rp).astype(np.float32)
    if fpa:
        sel = (sel - (counts.astype(np.float32) * fp + pp).astype(np.float32)).astype(np.float32)
    out[ids] = sel
    return out


def topk_indices(logits, k):
    """value descending, index ascending, -0.0 == +0.0 (rocm_kernels.rs:99-104 host_topk)."""
    v = np.asarray(logits, dtype=np.float32) + np.float32(0.0)      # -0.0 + 0.0 == +0.0
    order = np.lexsort((np.arange(v.size), -v.astype(np.float64)))
    return order[:k].astype(np.uint32)


def _gumbel_scores(scaled, u):
    return (scaled - np.log(-np.log(u.astype(np.float32))).astype(np.float32)).astype(np.float32)


def topp_mask(topk_logits, temperature, top_p):
    scaled = (np.asarray(topk_logits, np.float32) / np.float32(temperature)).astype(np.float32)
    e = np.exp(scaled - scaled.max())
    p = e / e.sum()
    c = np.cumsum(p)
    le = c <= top_p
    shift = np.concatenate([[False], le[:-1]])
    mask = le | shift
    if not mask.any():                 # reference leaves everything masked; crane_amd keeps the best token
        mask[0] = True
    return mask


def sample(logits, temperature=0.0, top_p=0.0, top_k=0, seed=299792458, draw=0, return_scores=False):
    logits = np.asarray(logits, dtype=np.float32)
    V = logits.size
    if not temperature > 0:
        return int(topk_indices(logits, 1)[0])
    top_p_active = 0 < top_p < 1
    k = top_k
    if k == 0 and top_p_active:
        k = 64
    k = min(k, 64, V)
    if 0 < k < V:
        idx = topk_indices(logits, k)
        tl = logits[idx]
        scaled = (tl / np.float32(temperature)).astype(np.float32)
        mask = topp_mask(tl, temperature, top_p) if top_p_active else np.ones(k, bool)
        u = uniform_stream(seed, draw, 64)[:k]
        sc = np.where(mask, _gumbel_scores(scaled, u), -np.inf)
        if return_scores:
            return idx, sc
        return int(idx[int(np.argmax(sc))])
    u = uniform_stream(seed, draw, V)
    sc = _gumbel_scores((logits / np.float32(temperature)).astype(np.float32), u)
    if return_scores:
        return np.arange(V), sc
    return int(np.argmax(sc))
