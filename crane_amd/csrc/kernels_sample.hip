// Device-side sampler: replaces the host path of crane-serve/src/engine/sampling.rs:169-373 (a [V] f32 D2H copy
// + ~24 ms host sort per token at V = 250 K, sampling.rs:25-27) and the RDNA-tuned top-k of
// crane-core/kernels/cuda/topk.cu with gfx950 kernels working on the logits already resident in HBM.
//   penalties : apply_penalties_inplace (sampling.rs:422-478): per distinct context token
//               logit = logit >= 0 ? logit / rp : logit * rp ;  logit -= count * freq + presence
//   top-k     : exact, total order "value descending, index ascending", -0.0 == +0.0
//               (topk.cu:70-81 key = order_preserving_u32(v) << 32 | ~idx); two-stage bitonic sort in LDS
//   sample    : sampling.rs:246-345: softmax(top-k logits / T), top-p mask (keep i if cumsum[i] <= p or
//               cumsum[i-1] <= p), Gumbel-max  argmax(logit / T - log(-log u)), u ~ U(1e-7, 0.999)  (:382-392)
// The uniform stream is a counter-based hash (seed, draw, lane); candle's StdRng stream is not reproduced -- the
// reference pins no sampled token anywhere (SURVEY 8c "parity unpinned").
#include "dev_common.h"
#include "kernels.h"

namespace cm {

__device__ __forceinline__ unsigned long long topk_key(float v, uint32_t idx) {
    if (v == 0.0f) v = 0.0f;                                   // -0.0 -> +0.0
    uint32_t b = __float_as_uint(v);
    b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);            // order-preserving map
    return ((unsigned long long)b << 32) | (unsigned long long)(~idx);
}

// descending bitonic sort of n (power of two) keys in LDS by one block
__device__ void bitonic_desc(unsigned long long* s, int n) {
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n; i += blockDim.x) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = s[i], b = s[ixj];
                    const bool up = (i & k) == 0;              // "up" blocks hold descending runs
                    if (up ? (a < b) : (a > b)) { s[i] = b; s[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
}

constexpr int TK_SLICE = 4096;

// stage 1: block b sorts logits[b*4096 .. +4096) and emits its kp best keys
__device__ __forceinline__ void topk_stage1_body(const float* __restrict__ logits, int n, int kp,
                                                 unsigned long long* __restrict__ cand, const uint32_t* __restrict__ sel) {
    __shared__ unsigned long long s[TK_SLICE];
    if (sel[2] != 0) return;                                   // the radix-select path already produced the answer
    const int base = blockIdx.x * TK_SLICE;
    for (int i = threadIdx.x; i < TK_SLICE; i += blockDim.x) {
        const int g = base + i;
        s[i] = g < n ? topk_key(logits[g], (uint32_t)g) : 0ull;   // 0 sorts below every real key
    }
    __syncthreads();
    bitonic_desc(s, TK_SLICE);
    for (int i = threadIdx.x; i < kp; i += blockDim.x) cand[(size_t)blockIdx.x * kp + i] = s[i];
}

// stage 2: one block folds all candidates, keeping the running best kp at the front
__device__ __forceinline__ void topk_stage2_body(const unsigned long long* __restrict__ cand, int ncand, int kp, int k,
                                                 uint32_t* __restrict__ idx_out, float* __restrict__ val_out,
                                                 const float* __restrict__ logits, const uint32_t* __restrict__ sel) {
    __shared__ unsigned long long s[TK_SLICE];
    if (sel[2] != 0) return;
    int consumed = 0;
    for (int i = threadIdx.x; i < TK_SLICE; i += blockDim.x) s[i] = 0ull;
    __syncthreads();
    bool first = true;
    while (consumed < ncand || first) {
        const int room = first ? TK_SLICE : TK_SLICE - kp;
        const int off = first ? 0 : kp;
        for (int i = threadIdx.x; i < room; i += blockDim.x) s[off + i] = (consumed + i < ncand) ? cand[consumed + i] : 0ull;
        consumed += room;
        first = false;
        __syncthreads();
        bitonic_desc(s, TK_SLICE);
    }
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        const uint32_t id = ~(uint32_t)(s[i] & 0xFFFFFFFFull);
        idx_out[i] = id;
        val_out[i] = logits[id];
    }
}

// ---------------------------------------------------------------------------------------------------------
// Fast path: radix select on the top 12 key bits, then sort only the survivors.
//   hist    : 4096-bin histogram of (ordered value bits >> 20), LDS atomics per 4096-logit slice, non-empty bins
//             flushed to HBM
//   select  : one block; suffix counts from the top bin; T = the bin holding the k-th largest value; every logit
//             in a bin >= T is a candidate (k <= n_cand); fast iff n_cand <= 4096 -- heavy exact ties (e.g. a
//             constant vector) fall back to the two-stage bitonic kernels above, which early-exit otherwise
//   collect : candidates appended in any order (the sort restores the total order)
//   final   : one block sorts pow2(n_cand) keys (typically 64..256) and emits the first k
// sel = {T, n_cand, fast, counter}
// ---------------------------------------------------------------------------------------------------------
constexpr int TK_BINS = 4096, TK_CAP = 4096;

__device__ __forceinline__ uint32_t topk_ord(float v) {
    if (v == 0.0f) v = 0.0f;
    const uint32_t b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__device__ __forceinline__ void topk_hist_body(const float* __restrict__ logits, int n, uint32_t* __restrict__ hist) {
    __shared__ uint32_t h[TK_BINS];
    for (int i = threadIdx.x; i < TK_BINS; i += blockDim.x) h[i] = 0;
    __syncthreads();
    const int base = blockIdx.x * TK_SLICE;
    for (int i = threadIdx.x; i < TK_SLICE; i += blockDim.x) {
        const int g = base + i;
        if (g < n) atomicAdd(&h[topk_ord(logits[g]) >> 20], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < TK_BINS; i += blockDim.x)
        if (h[i]) atomicAdd(&hist[i], h[i]);
}

__device__ __forceinline__ void topk_select_body(uint32_t* __restrict__ hist, int k, uint32_t* __restrict__ sel) {
    __shared__ uint32_t part[1024];
    const int t = threadIdx.x;
    uint32_t c[4], sum = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { c[j] = hist[4 * t + j]; hist[4 * t + j] = 0; sum += c[j]; }      // zeroed for the next call
    part[t] = sum;
    __syncthreads();
    // suffix sums over the 1024 partials (Hillis-Steele, 10 steps)
    for (int d = 1; d < 1024; d <<= 1) {
        const uint32_t v = (t + d < 1024) ? part[t + d] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t above = (t + 1 < 1024) ? part[t + 1] : 0u;          // count in bins > 4t+3
#pragma unroll
    for (int j = 3; j >= 0; --j) {
        const uint32_t incl = above + c[j];
        if (above < (uint32_t)k && incl >= (uint32_t)k) {
            sel[0] = (uint32_t)(4 * t + j);
            sel[1] = incl;
            sel[2] = incl <= (uint32_t)TK_CAP ? 1u : 0u;
            sel[3] = 0u;
        }
        above = incl;
    }
}

__device__ __forceinline__ void topk_collect_body(const float* __restrict__ logits, int n, uint32_t* __restrict__ sel,
                                                  unsigned long long* __restrict__ cand) {
    if (sel[2] == 0) return;
    const uint32_t T = sel[0];
    const int base = blockIdx.x * TK_SLICE;
    for (int i = threadIdx.x; i < TK_SLICE; i += blockDim.x) {
        const int g = base + i;
        if (g >= n) break;
        const float v = logits[g];
        if ((topk_ord(v) >> 20) >= T) {
            const uint32_t p = atomicAdd(&sel[3], 1u);
            if (p < (uint32_t)TK_CAP) cand[p] = topk_key(v, (uint32_t)g);
        }
    }
}

__device__ __forceinline__ void topk_final_body(const unsigned long long* __restrict__ cand, const uint32_t* __restrict__ sel,
                                                int k, uint32_t* __restrict__ idx_out, float* __restrict__ val_out,
                                                const float* __restrict__ logits) {
    __shared__ unsigned long long s[TK_CAP];
    if (sel[2] == 0) return;
    const int n = (int)sel[1];
    int P = 64;
    while (P < n) P <<= 1;
    for (int i = threadIdx.x; i < P; i += blockDim.x) s[i] = i < n ? cand[i] : 0ull;
    __syncthreads();
    bitonic_desc(s, P);
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        const uint32_t id = ~(uint32_t)(s[i] & 0xFFFFFFFFull);
        idx_out[i] = id;
        val_out[i] = logits[id];
    }
}

__device__ __forceinline__ void penalties_body(float* __restrict__ logits, const uint32_t* __restrict__ ids, const uint32_t* __restrict__ counts,
                                               int n, float rp, float rp_inv, int true_div, float fp, float pp, int V) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t t = ids[i];
    if (t >= (uint32_t)V) return;
    float v = logits[t];
    // sampling.rs:452 divides through candle's `Tensor / f64` == multiply by (f32)(1/rp); the single-request
    // generate() path (candle_transformers apply_repeat_penalty, model.rs:306-315) uses a true division
    if (rp != 1.0f) v = v >= 0.f ? (true_div ? __fdiv_rn(v, rp) : __fmul_rn(v, rp_inv)) : __fmul_rn(v, rp);
    if (fp != 0.0f || pp != 0.0f) v = __fsub_rn(v, __fadd_rn(__fmul_rn((float)counts[i], fp), pp));
    logits[t] = v;
}

__device__ __forceinline__ float uniform_open(uint32_t seed_lo, uint32_t seed_hi, uint32_t draw, uint32_t i) {
    uint32_t h = fmix32(seed_lo ^ (i * 0x9E3779B1u));
    h = fmix32(h ^ seed_hi ^ (draw * 0x85EBCA6Bu));
    const float u01 = (float)(h >> 8) * (1.0f / 16777216.0f);          // [0, 1)
    return 1e-7f + u01 * (0.999f - 1e-7f);                             // rand_like(1e-7, 0.999)
}

// one wave: k <= 64 candidates (sorted, best first)
__device__ __forceinline__ void sample_topk_body(const uint32_t* __restrict__ idx, const float* __restrict__ val, int k,
                                                 float temperature, float top_p, uint32_t seed_lo, uint32_t seed_hi,
                                                 uint32_t draw, uint32_t* __restrict__ token_out) {
    const int lane = threadIdx.x;
    const bool act = lane < k;
    const float lg = act ? val[lane] : -INFINITY;
    const float scaled = act ? lg / temperature : -INFINITY;
    bool keep = act;
    if (top_p > 0.f && top_p < 1.f) {
        const float mx = wave_max(scaled);
        const float e = act ? expf(scaled - mx) : 0.f;
        const float sum = wave_sum(e);
        const float p = e / sum;
        float c = p;                                                   // inclusive prefix sum over the lanes
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const float t = __shfl_up(c, d);
            if (lane >= d) c += t;
        }
        const float prev = __shfl_up(c, 1);
        keep = act && ((c <= top_p) || (lane > 0 && prev <= top_p));
        if (__ballot(keep) == 0ull) keep = (lane == 0);               // degenerate p < p_0: keep the best token
    }
    const float u = uniform_open(seed_lo, seed_hi, draw, (uint32_t)lane);
    float score = keep ? scaled - logf(-logf(u)) : -INFINITY;
    // arg-max over the wave, lowest lane wins ties
    float best = score; int bl = lane;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const float ob = __shfl_xor(best, d); const int ol = __shfl_xor(bl, d);
        if (ob > best || (ob == best && ol < bl)) { best = ob; bl = ol; }
    }
    if (lane == 0) token_out[0] = idx[bl];
}

// full-vocabulary Gumbel-max (top_k == 0 and no top_p): phase 1 per block, phase 2 = gumbel_final_kernel
__global__ __launch_bounds__(256) void gumbel_full_kernel(const float* __restrict__ logits, int V, float temperature, uint32_t seed_lo,
                                                          uint32_t seed_hi, uint32_t draw, float* __restrict__ pmax,
                                                          int* __restrict__ pidx) {
    __shared__ float sm[256];
    __shared__ int si[256];
    float b = -INFINITY; int bi = 0x7FFFFFFF;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < V; i += gridDim.x * 256) {
        const float s = logits[i] / temperature - logf(-logf(uniform_open(seed_lo, seed_hi, draw, (uint32_t)i)));
        if (s > b || (s == b && i < bi)) { b = s; bi = i; }
    }
    sm[threadIdx.x] = b; si[threadIdx.x] = bi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            const float v = sm[threadIdx.x + s]; const int ix = si[threadIdx.x + s];
            if (v > sm[threadIdx.x] || (v == sm[threadIdx.x] && ix < si[threadIdx.x])) { sm[threadIdx.x] = v; si[threadIdx.x] = ix; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { pmax[blockIdx.x] = sm[0]; pidx[blockIdx.x] = si[0]; }
}

__global__ __launch_bounds__(256) void gumbel_final_kernel(const float* __restrict__ pmax, const int* __restrict__ pidx, int n,
                                                           uint32_t* __restrict__ token_out) {
    __shared__ float sm[256];
    __shared__ int si[256];
    float b = -INFINITY; int bi = 0x7FFFFFFF;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float v = pmax[i]; const int ix = pidx[i];
        if (v > b || (v == b && ix < bi)) { b = v; bi = ix; }
    }
    sm[threadIdx.x] = b; si[threadIdx.x] = bi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            const float v = sm[threadIdx.x + s]; const int ix = si[threadIdx.x + s];
            if (v > sm[threadIdx.x] || (v == sm[threadIdx.x] && ix < si[threadIdx.x])) { sm[threadIdx.x] = v; si[threadIdx.x] = ix; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) token_out[0] = (uint32_t)si[0];
}

// ---- the kernels: one row (pointers as arguments) or a table of rows (blockIdx.y = row; every row has its own scratch) ----
__global__ __launch_bounds__(1024) void topk_stage1_kernel(const float* logits, int n, int kp, unsigned long long* cand, const uint32_t* sel) {
    topk_stage1_body(logits, n, kp, cand, sel);
}
__global__ __launch_bounds__(1024) void topk_stage2_kernel(const unsigned long long* cand, int ncand, int kp, int k, uint32_t* idx_out,
                                                           float* val_out, const float* logits, const uint32_t* sel) {
    topk_stage2_body(cand, ncand, kp, k, idx_out, val_out, logits, sel);
}
__global__ __launch_bounds__(1024) void topk_hist_kernel(const float* logits, int n, uint32_t* hist) { topk_hist_body(logits, n, hist); }
__global__ __launch_bounds__(1024) void topk_select_kernel(uint32_t* hist, int k, uint32_t* sel) { topk_select_body(hist, k, sel); }
__global__ __launch_bounds__(1024) void topk_collect_kernel(const float* logits, int n, uint32_t* sel, unsigned long long* cand) {
    topk_collect_body(logits, n, sel, cand);
}
__global__ __launch_bounds__(1024) void topk_final_kernel(const unsigned long long* cand, const uint32_t* sel, int k, uint32_t* idx_out,
                                                          float* val_out, const float* logits) {
    topk_final_body(cand, sel, k, idx_out, val_out, logits);
}
__global__ void penalties_kernel(float* logits, const uint32_t* ids, const uint32_t* counts, int n, float rp, float rp_inv, int true_div,
                                 float fp, float pp, int V) {
    penalties_body(logits, ids, counts, n, rp, rp_inv, true_div, fp, pp, V);
}
__global__ __launch_bounds__(64) void sample_topk_kernel(const uint32_t* idx, const float* val, int k, float temperature, float top_p,
                                                         uint32_t seed_lo, uint32_t seed_hi, uint32_t draw, uint32_t* token_out) {
    sample_topk_body(idx, val, k, temperature, top_p, seed_lo, seed_hi, draw, token_out);
}

__global__ __launch_bounds__(1024) void topk_stage1_rows_kernel(const SampleRowDev* tab, int n) {
    const SampleRowDev& r = tab[blockIdx.y];
    topk_stage1_body(r.logits, n, r.kp, r.cand, r.sel);
}
__global__ __launch_bounds__(1024) void topk_stage2_rows_kernel(const SampleRowDev* tab, int nb) {
    const SampleRowDev& r = tab[blockIdx.y];
    topk_stage2_body(r.cand, nb * r.kp, r.kp, r.k, r.idx_out, r.val, r.logits, r.sel);
}
__global__ __launch_bounds__(1024) void topk_hist_rows_kernel(const SampleRowDev* tab, int n) {
    const SampleRowDev& r = tab[blockIdx.y];
    topk_hist_body(r.logits, n, r.hist);
}
__global__ __launch_bounds__(1024) void topk_select_rows_kernel(const SampleRowDev* tab) {
    const SampleRowDev& r = tab[blockIdx.y];
    topk_select_body(r.hist, r.k, r.sel);
}
__global__ __launch_bounds__(1024) void topk_collect_rows_kernel(const SampleRowDev* tab, int n) {
    const SampleRowDev& r = tab[blockIdx.y];
    topk_collect_body(r.logits, n, r.sel, r.cand);
}
__global__ __launch_bounds__(1024) void topk_final_rows_kernel(const SampleRowDev* tab) {
    const SampleRowDev& r = tab[blockIdx.y];
    topk_final_body(r.cand, r.sel, r.k, r.idx_out, r.val, r.logits);
}
__global__ void penalties_rows_kernel(const SampleRowDev* tab, int V) {
    const SampleRowDev& r = tab[blockIdx.y];
    penalties_body(r.logits, r.pen_ids, r.pen_counts, r.pen_n, r.rp, r.rp_inv, r.true_div, r.fp, r.pp, V);
}
__global__ __launch_bounds__(64) void sample_topk_rows_kernel(const SampleRowDev* tab) {
    const SampleRowDev& r = tab[blockIdx.y];
    if (r.sample) sample_topk_body(r.idx_out, r.val, r.k, r.temperature, r.top_p, r.seed_lo, r.seed_hi, r.draw, r.tok);
}

int topk_pad(int k) { int p = 1; while (p < k) p <<= 1; return p; }
int topk_blocks(int n) { return (n + TK_SLICE - 1) / TK_SLICE; }

// cand: >= max(topk_blocks(n) * topk_pad(k), 4096) keys; hist: 4096 zeroed u32 (left zeroed); sel: 4 u32
void launch_topk(const float* logits, int n, int k, unsigned long long* cand, uint32_t* hist, uint32_t* sel, uint32_t* idx_out,
                 float* val_out, hipStream_t s) {
    const int kp = topk_pad(k), nb = topk_blocks(n);
    hipLaunchKernelGGL(topk_hist_kernel, dim3(nb), dim3(1024), 0, s, logits, n, hist);
    hipLaunchKernelGGL(topk_select_kernel, dim3(1), dim3(1024), 0, s, hist, k, sel);
    hipLaunchKernelGGL(topk_collect_kernel, dim3(nb), dim3(1024), 0, s, logits, n, sel, cand);
    hipLaunchKernelGGL(topk_final_kernel, dim3(1), dim3(1024), 0, s, cand, sel, k, idx_out, val_out, logits);
    // exact-tie fallback (early-exits when the fast path ran)
    hipLaunchKernelGGL(topk_stage1_kernel, dim3(nb), dim3(1024), 0, s, logits, n, kp, cand, sel);
    hipLaunchKernelGGL(topk_stage2_kernel, dim3(1), dim3(1024), 0, s, cand, nb * kp, kp, k, idx_out, val_out, logits, sel);
}
void launch_penalties(float* logits, const uint32_t* ids, const uint32_t* counts, int n, float rp, bool true_div, float fp, float pp,
                      int V, hipStream_t s) {
    if (n <= 0) return;
    const float rp_inv = (float)(1.0 / (double)rp);
    hipLaunchKernelGGL(penalties_kernel, dim3((n + 255) / 256), dim3(256), 0, s, logits, ids, counts, n, rp, rp_inv, true_div ? 1 : 0, fp,
                       pp, V);
}
void launch_sample_topk(const uint32_t* idx, const float* val, int k, float temperature, float top_p, uint64_t seed, uint32_t draw,
                        uint32_t* token_out, hipStream_t s) {
    hipLaunchKernelGGL(sample_topk_kernel, dim3(1), dim3(64), 0, s, idx, val, k, temperature, top_p, (uint32_t)seed,
                       (uint32_t)(seed >> 32), draw, token_out);
}
void launch_gumbel_full(const float* logits, int V, float temperature, uint64_t seed, uint32_t draw, float* pmax, int* pidx, int blocks,
                        uint32_t* token_out, hipStream_t s) {
    hipLaunchKernelGGL(gumbel_full_kernel, dim3(blocks), dim3(256), 0, s, logits, V, temperature, (uint32_t)seed, (uint32_t)(seed >> 32),
                       draw, pmax, pidx);
    hipLaunchKernelGGL(gumbel_final_kernel, dim3(1), dim3(256), 0, s, pmax, pidx, blocks, token_out);
}

// every row of a decode group in the same eight launches (the per-row path costs eight launches and a copy PER ROW: 128 rows x 9
// stream operations of ~3 us each were a quarter of a 128-sequence engine round)
void launch_sample_rows(const SampleRowDev* tab, int nrows, int V, int max_pen, hipStream_t s) {
    const int nb = topk_blocks(V);
    if (max_pen > 0) hipLaunchKernelGGL(penalties_rows_kernel, dim3((max_pen + 255) / 256, nrows), dim3(256), 0, s, tab, V);
    hipLaunchKernelGGL(topk_hist_rows_kernel, dim3(nb, nrows), dim3(1024), 0, s, tab, V);
    hipLaunchKernelGGL(topk_select_rows_kernel, dim3(1, nrows), dim3(1024), 0, s, tab);
    hipLaunchKernelGGL(topk_collect_rows_kernel, dim3(nb, nrows), dim3(1024), 0, s, tab, V);
    hipLaunchKernelGGL(topk_final_rows_kernel, dim3(1, nrows), dim3(1024), 0, s, tab);
    hipLaunchKernelGGL(topk_stage1_rows_kernel, dim3(nb, nrows), dim3(1024), 0, s, tab, V);
    hipLaunchKernelGGL(topk_stage2_rows_kernel, dim3(1, nrows), dim3(1024), 0, s, tab, nb);
    hipLaunchKernelGGL(sample_topk_rows_kernel, dim3(1, nrows), dim3(64), 0, s, tab);
}

}  // namespace cm
