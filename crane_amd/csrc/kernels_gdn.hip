// Gated Delta Net (Qwen 3.5 / 3.6 / 3.8 linear-attention layers) -- decode step and sequential prefill.
//
// One kernel per layer does everything between the input projection GEMV/GEMM and the output
// projection (reference: ops/gdn/layer.rs:122-182 = ~12 candle launches for the conv alone,
// ops/gdn/conv.rs:63-73):
//   causal depthwise conv1d (k = 4) + SiLU with rolling state        ops/gdn/conv.rs:23-101
//   split q|k|v, key-head -> value-head expansion (HF "Interleaved")  ops/gdn/layer.rs:184-238
//   L2-norm(q), L2-norm(k) (eps 1e-6), q * 1/sqrt(K)                   ops/gdn/backend.rs:26-56,100-108
//   beta = sigmoid(b), g = -exp(A_log) * softplus(a + dt_bias)         ops/gdn/backend.rs:197-211
//   S *= exp(g); kv = S^T k; delta = (v - kv) beta; S += k (x) delta; y = S^T q   backend.rs:90-156,
//                                                                      kernels/cuda/gdn.cu:83-124
//   RmsNormGated: rms_norm(y, w) * silu(z)  (plain weight)             ops/gdn/norm.rs:39-45
// State is f32 (ops/gdn/cache.rs:12-13).  HBM-bound at decode: 2 * K * V * 4 B per value head per
// token (read + write of the state), 0.5 flop/byte.
//
// Layout: one block per value head, one thread per state COLUMN v (V = 128 threads); the column's
// K = 128 state values live in registers across the two passes (kv needs all of k before the
// update); S[k][v] is stored [K][V] so a wave reads/writes 256 contiguous bytes per k.
// Conv state is double-buffered by position parity so that the blocks of one key-head group (which
// share q/k channels) never read a window another block is rolling.
#include "dev_common.h"
#include "kernels.h"

namespace cm {

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }

// grid = NV value heads, block = 128 threads.  Processes S tokens sequentially (S = 1 at decode).
__global__ __launch_bounds__(128) void gdn_kernel(GdnArgs a) {
    constexpr int K = 128, V = 128, KER = 4;
    __shared__ float qs[K], ks[K];
    __shared__ float red[8];
    const int h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bq = blockIdx.y;                                 // sequence of a batched decode step (0 otherwise)
    const int nk = a.NV / a.vpg;
    const int kh = a.chunked ? h % nk : h / a.vpg;             // Interleaved (HF): h / vpg; Chunked (GGUF): h % NK
    const int cq = kh * K + tid, ck = a.key_dim + kh * K + tid, cv = 2 * a.key_dim + h * V + tid;
    const int conv_dim = 2 * a.key_dim + a.NV * V;
    const int proj_stride = a.proj_stride;
    const bool writer = a.chunked ? (h < nk) : (h % a.vpg) == 0;   // one block per key head rolls the q/k windows

    // conv weights of my three channels, state column in registers
    float wq[KER], wk[KER], wv[KER];
#pragma unroll
    for (int j = 0; j < KER; ++j) { wq[j] = a.conv_w[cq * KER + j]; wk[j] = a.conv_w[ck * KER + j]; wv[j] = a.conv_w[cv * KER + j]; }
    const int start_pos = a.st ? a.st[bq].pos : a.start_pos;
    const int slot = a.st ? a.st[bq].slot : a.slot;
    float* conv_state = a.conv_pool + ((size_t)slot * a.gdn_layers + a.layer_idx) * 2 * conv_dim * (KER - 1);
    const int par_in = start_pos & 1;
    const float* cs_in = conv_state + (size_t)par_in * conv_dim * (KER - 1);
    float hq[KER - 1], hk[KER - 1], hv[KER - 1];
#pragma unroll
    for (int j = 0; j < KER - 1; ++j) {
        hq[j] = cs_in[cq * (KER - 1) + j]; hk[j] = cs_in[ck * (KER - 1) + j]; hv[j] = cs_in[cv * (KER - 1) + j];
    }
    float* Sg = a.state_pool + (((size_t)slot * a.gdn_layers + a.layer_idx) * a.NV + h) * K * V;
    float S[K];
#pragma unroll
    for (int k = 0; k < K; ++k) S[k] = Sg[k * V + tid];
    const float neg_exp_a = -expf(a.A_log[h]);
    const float dtb = a.dt_bias[h];
    const float gw = a.gnorm_w[tid];
    const float qscale = 0.08838834764831845f;                 // 1/sqrt(128)

    for (int t = 0; t < a.S; ++t) {
        const float* pr = a.proj + (size_t)bq * a.batch_proj_stride + (size_t)t * proj_stride;
        // ---- conv + SiLU; roll the windows ----
        const float xq = pr[cq], xk = pr[ck], xv = pr[cv];
        float q = hq[0] * wq[0] + hq[1] * wq[1] + hq[2] * wq[2] + xq * wq[3];
        float k = hk[0] * wk[0] + hk[1] * wk[1] + hk[2] * wk[2] + xk * wk[3];
        float v = hv[0] * wv[0] + hv[1] * wv[1] + hv[2] * wv[2] + xv * wv[3];
        hq[0] = hq[1]; hq[1] = hq[2]; hq[2] = xq;
        hk[0] = hk[1]; hk[1] = hk[2]; hk[2] = xk;
        hv[0] = hv[1]; hv[1] = hv[2]; hv[2] = xv;
        q = silu_f(q); k = silu_f(k); v = silu_f(v);
        // ---- L2 norms over the key head (128 threads = 2 waves) ----
        float sq = wave_sum(q * q), sk = wave_sum(k * k);
        __syncthreads();                                        // previous iteration done with qs/ks/red
        if (lane == 0) { red[wave] = sq; red[2 + wave] = sk; }
        __syncthreads();
        q = q / sqrtf(red[0] + red[1] + 1e-6f) * qscale;
        k = k / sqrtf(red[2] + red[3] + 1e-6f);
        qs[tid] = q; ks[tid] = k;
        // ---- gates ----
        const float beta = 1.0f / (1.0f + expf(-pr[conv_dim + a.NV * V + h]));
        const float av = pr[conv_dim + a.NV * V + a.NV + h] + dtb;
        const float g = neg_exp_a * logf(1.0f + expf(av));
        const float decay = expf(g);
        __syncthreads();
        // ---- recurrence on my column ----
        float kv4[4] = {0.f, 0.f, 0.f, 0.f};                     // 4 independent chains instead of one 128-deep dependent one
#pragma unroll
        for (int kk = 0; kk < K; ++kk) { S[kk] *= decay; kv4[kk & 3] += S[kk] * ks[kk]; }
        const float kv = (kv4[0] + kv4[1]) + (kv4[2] + kv4[3]);
        const float delta = (v - kv) * beta;
        float y4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < K; ++kk) { S[kk] += ks[kk] * delta; y4[kk & 3] += S[kk] * qs[kk]; }
        const float y = (y4[0] + y4[1]) + (y4[2] + y4[3]);
        // ---- gated RMSNorm over the value head ----
        const float sy = wave_sum(y * y);
        if (lane == 0) red[4 + wave] = sy;
        __syncthreads();
        const float rms = 1.0f / sqrtf((red[4] + red[5]) / (float)V + a.eps);
        const float z = pr[conv_dim + h * V + tid];
        a.out[(size_t)bq * a.batch_out_stride + (size_t)t * a.out_stride + h * V + tid] = y * rms * gw * silu_f(z);
    }
    // ---- write back state and conv windows (other parity buffer) ----
#pragma unroll
    for (int k = 0; k < K; ++k) Sg[k * V + tid] = S[k];
    float* cs_out = conv_state + (size_t)((start_pos + a.S) & 1) * conv_dim * (KER - 1);
#pragma unroll
    for (int j = 0; j < KER - 1; ++j) {
        if (writer) { cs_out[cq * (KER - 1) + j] = hq[j]; cs_out[ck * (KER - 1) + j] = hk[j]; }
        cs_out[cv * (KER - 1) + j] = hv[j];
    }
}

// ---------------------------------------------------------------------------------------------------------
// Decode step on NV x 4 workgroups.  The fused kernel above reads and writes a value head's 64 KiB state from ONE
// workgroup: 16 (Qwen3.5-0.8B) ... 48 (Qwen3.8-27B) of 256 CUs stream the only HBM-bound bytes of the layer (measured
// 8.7 us per layer on the 0.8B model = 0.24 TB/s).  Here a value head is four workgroups of 32 state columns each
// (one column = 256 threads / 32: 8 k-slices of 16 state values per thread; a wave reads 2 x 128 contiguous bytes per k).
// Everything per token is recomputed by each of them (conv + SiLU + L2 norm of the key head's q / k: 256 channels), the
// recurrence runs on the workgroup's own columns, and the gated RMSNorm -- which needs the head's 128 y values -- is done
// by the LAST of the four workgroups to arrive: raw y and the partial sums of squares go out as write-through (sc1)
// stores, every wave drains them, one lane takes a ticket; the workgroup that draws 3 reads them back with sc1 loads
// (no fences, no polling: MI355X_MICROARCH.md "handoff-flag", Guideline 16 R1).  Same arithmetic as gdn_kernel except
// for the association of the two 128-term sums over k (8 partial sums of 16 terms).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gdn_decode_kernel(GdnArgs a) {
    constexpr int K = 128, V = 128, KER = 4, CB = 32, KS = 8, KP = K / KS;
    __shared__ float qs[K], ks_[K], vs[CB];
    __shared__ float red[8];
    __shared__ float part[KS][CB];
    __shared__ float bd[2];
    __shared__ int last_flag;
    const int h = blockIdx.x, cb = blockIdx.y, bq = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = tid & (CB - 1), ksl = tid / CB;              // column inside the block, k-slice
    const int v = cb * CB + c;
    const int nk = a.NV / a.vpg;
    const int kh = a.chunked ? h % nk : h / a.vpg;
    const int conv_dim = 2 * a.key_dim + a.NV * V;
    const bool qk_writer = cb == 0 && (a.chunked ? (h < nk) : (h % a.vpg) == 0);
    const int start_pos = a.st[bq].pos, slot = a.st[bq].slot;
    float* conv_state = a.conv_pool + ((size_t)slot * a.gdn_layers + a.layer_idx) * 2 * conv_dim * (KER - 1);
    const float* cs_in = conv_state + (size_t)(start_pos & 1) * conv_dim * (KER - 1);
    float* cs_out = conv_state + (size_t)((start_pos + 1) & 1) * conv_dim * (KER - 1);
    const float* pr = a.proj + (size_t)bq * a.batch_proj_stride;
    // state slice: requested first (the only HBM-bound bytes)
    float* Sg = a.state_pool + (((size_t)slot * a.gdn_layers + a.layer_idx) * a.NV + h) * K * V;
    float S[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) S[k] = __builtin_nontemporal_load(&Sg[(ksl * KP + k) * V + v]);
    // conv + SiLU: threads 0..127 the q channel, 128..255 the k channel of the key head; threads < 32 also this block's v columns
    {
        const int ch = (tid < K ? kh * K + tid : a.key_dim + kh * K + (tid - K));
        const float w0 = a.conv_w[ch * KER], w1 = a.conv_w[ch * KER + 1], w2 = a.conv_w[ch * KER + 2], w3 = a.conv_w[ch * KER + 3];
        const float h0 = cs_in[ch * (KER - 1)], h1 = cs_in[ch * (KER - 1) + 1], h2 = cs_in[ch * (KER - 1) + 2];
        const float x = pr[ch];
        const float y = silu_f(h0 * w0 + h1 * w1 + h2 * w2 + x * w3);
        if (qk_writer) { cs_out[ch * (KER - 1)] = h1; cs_out[ch * (KER - 1) + 1] = h2; cs_out[ch * (KER - 1) + 2] = x; }
        const float ss = wave_sum(y * y);                      // waves 0,1 = q; 2,3 = k
        if (lane == 0) red[wave] = ss;
        if (tid < K) qs[tid] = y; else ks_[tid - K] = y;
        if (tid < CB) {
            const int cv = 2 * a.key_dim + h * V + v;          // v == cb * CB + tid here
            const float u0 = a.conv_w[cv * KER], u1 = a.conv_w[cv * KER + 1], u2 = a.conv_w[cv * KER + 2], u3 = a.conv_w[cv * KER + 3];
            const float g0 = cs_in[cv * (KER - 1)], g1 = cs_in[cv * (KER - 1) + 1], g2 = cs_in[cv * (KER - 1) + 2];
            const float xv = pr[cv];
            vs[tid] = silu_f(g0 * u0 + g1 * u1 + g2 * u2 + xv * u3);
            cs_out[cv * (KER - 1)] = g1; cs_out[cv * (KER - 1) + 1] = g2; cs_out[cv * (KER - 1) + 2] = xv;
        }
        if (tid == 0) {
            const float beta = 1.0f / (1.0f + expf(-pr[conv_dim + a.NV * V + h]));
            const float av = pr[conv_dim + a.NV * V + a.NV + h] + a.dt_bias[h];
            bd[0] = beta;
            bd[1] = expf(-expf(a.A_log[h]) * logf(1.0f + expf(av)));
        }
    }
    __syncthreads();
    const float dq = sqrtf(red[0] + red[1] + 1e-6f), dk = sqrtf(red[2] + red[3] + 1e-6f);
    const float beta = bd[0], decay = bd[1];
    float kf[KP], qf[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) { kf[k] = ks_[ksl * KP + k] / dk; qf[k] = qs[ksl * KP + k] / dq * 0.08838834764831845f; }   // 1/sqrt(128)
    // ---- recurrence on this thread's 16 state values of column v ----
    float kv2[2] = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < KP; ++k) { S[k] *= decay; kv2[k & 1] += S[k] * kf[k]; }
    part[ksl][c] = kv2[0] + kv2[1];
    __syncthreads();
    float kv = 0.f;
#pragma unroll
    for (int j = 0; j < KS; ++j) kv += part[j][c];
    const float delta = (vs[c] - kv) * beta;
    float y2[2] = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < KP; ++k) { S[k] += kf[k] * delta; y2[k & 1] += S[k] * qf[k]; }
#pragma unroll
    for (int k = 0; k < KP; ++k) __builtin_nontemporal_store(S[k], &Sg[(ksl * KP + k) * V + v]);
    __syncthreads();                                           // everyone has read part[][] (kv)
    part[ksl][c] = y2[0] + y2[1];
    __syncthreads();
    if (a.defer_norm) {          // single-sequence decode: the out_proj GEMV normalises while it stages x (PRO_GDNNORM)
        if (tid < CB) {
            float y = 0.f;
#pragma unroll
            for (int j = 0; j < KS; ++j) y += part[j][tid];
            a.out[(size_t)bq * a.batch_out_stride + h * V + cb * CB + tid] = y;
        }
        return;
    }
    // ---- raw y of the block's 32 columns + partial sum of squares -> write-through; ticket; last arriver normalises ----
    float* yraw = a.gdn_scratch + ((size_t)bq * a.NV + h) * (V + 4);
    int* ticket = a.gdn_ticket + (size_t)bq * a.NV + h;
    if (tid < CB) {
        float y = 0.f;
#pragma unroll
        for (int j = 0; j < KS; ++j) y += part[j][tid];
        __hip_atomic_store(&yraw[cb * CB + tid], y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        float sy = y * y;
        sy += __shfl_xor(sy, 16); sy += __shfl_xor(sy, 8); sy += __shfl_xor(sy, 4); sy += __shfl_xor(sy, 2); sy += __shfl_xor(sy, 1);
        if (tid == 0) __hip_atomic_store(&yraw[V + cb], sy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // every storing wave drains its write-through stores
    __syncthreads();
    if (tid == 0) last_flag = (atomicAdd(ticket, 1) == (int)gridDim.y - 1) ? 1 : 0;
    __syncthreads();
    if (!last_flag) return;
    if (tid < V) {
        const float y = __hip_atomic_load(&yraw[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        float tot = 0.f;
        for (int j = 0; j < (int)gridDim.y; ++j) tot += __hip_atomic_load(&yraw[V + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float rms = 1.0f / sqrtf(tot / (float)V + a.eps);
        const float z = pr[conv_dim + h * V + tid];
        a.out[(size_t)bq * a.batch_out_stride + h * V + tid] = y * rms * a.gnorm_w[tid] * silu_f(z);
    }
    if (tid == 0) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // next launch starts from 0
}

// ---------------------------------------------------------------------------------------------------------
// Prefill in three passes.  The fused kernel above walks a prompt with 4 barriers, two block reductions and a handful
// of transcendental functions per token on ONE block per value head (~5 us per token and layer).  Everything except
// the state recurrence is independent across tokens, so:
//   gdn_pre_kernel  (grid S x (NK + NV)): conv + SiLU, L2 norms, q scale, beta, exp(g)  -> q^ k^ v beta decay per token
//   gdn_scan_kernel (grid NV x 4): the recurrence only; 16 lanes share a state column (8 of the 128 k each, partial
//                   sums folded with one DPP row reduction), q^ / k^ of 8 tokens staged per barrier
//   gdn_post_kernel (grid S x NV): gated RMSNorm of the raw y rows in place
// Same per-token arithmetic as the fused kernel except for the association of the two 128-term sums over k.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void gdn_pre_kernel(GdnArgs a) {
    constexpr int K = 128, V = 128, KER = 4;
    __shared__ float red[4];
    const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nk = a.NV / a.vpg;
    const int conv_dim = 2 * a.key_dim + a.NV * V;
    const float* conv_state = a.conv_pool + ((size_t)a.slot * a.gdn_layers + a.layer_idx) * 2 * conv_dim * (KER - 1);
    const float* cs_in = conv_state + (size_t)(a.start_pos & 1) * conv_dim * (KER - 1);
    float* cs_out = a.conv_pool + ((size_t)a.slot * a.gdn_layers + a.layer_idx) * 2 * conv_dim * (KER - 1) +
                    (size_t)((a.start_pos + a.S) & 1) * conv_dim * (KER - 1);
    // input of channel c at token tt (tt < 0: the rolling window left by the previous call, oldest first)
    auto xin = [&](int c, int tt) -> float {
        return tt >= 0 ? a.proj[(size_t)tt * a.proj_stride + c] : cs_in[c * (KER - 1) + (KER - 1) + tt];
    };
    auto conv = [&](int c) -> float {
        const float x0 = xin(c, t - 3), x1 = xin(c, t - 2), x2 = xin(c, t - 1), x3 = xin(c, t);
        const float* w = a.conv_w + c * KER;
        return silu_f(x0 * w[0] + x1 * w[1] + x2 * w[2] + x3 * w[3]);
    };
    auto roll = [&](int c) {                         // the last token's block leaves the window for the next call
        if (t == a.S - 1) {
#pragma unroll
            for (int j = 0; j < KER - 1; ++j) cs_out[c * (KER - 1) + j] = xin(c, a.S - (KER - 1) + j);
        }
    };
    const float* pr = a.proj + (size_t)t * a.proj_stride;
    if ((int)blockIdx.y < nk) {                      // q and k of key head kh
        const int kh = blockIdx.y, cq = kh * K + tid, ck = a.key_dim + kh * K + tid;
        float q = conv(cq), k = conv(ck);
        const float sq = wave_sum(q * q), sk = wave_sum(k * k);
        if (lane == 0) { red[wave] = sq; red[2 + wave] = sk; }
        __syncthreads();
        q = q / sqrtf(red[0] + red[1] + 1e-6f) * 0.08838834764831845f;      // 1/sqrt(128)
        k = k / sqrtf(red[2] + red[3] + 1e-6f);
        a.pre_q[(size_t)t * a.key_dim + cq] = q;
        a.pre_k[(size_t)t * a.key_dim + kh * K + tid] = k;
        roll(cq); roll(ck);
    } else {                                         // v, beta, decay of value head h
        const int h = blockIdx.y - nk, cv = 2 * a.key_dim + h * V + tid;
        a.pre_v[((size_t)t * a.NV + h) * V + tid] = conv(cv);
        if (tid == 0) {
            const float beta = 1.0f / (1.0f + expf(-pr[conv_dim + a.NV * V + h]));
            const float av = pr[conv_dim + a.NV * V + a.NV + h] + a.dt_bias[h];
            const float g = -expf(a.A_log[h]) * logf(1.0f + expf(av));
            a.pre_bd[((size_t)t * a.NV + h) * 2] = beta;
            a.pre_bd[((size_t)t * a.NV + h) * 2 + 1] = expf(g);
        }
        roll(cv);
    }
}

// grid = (NV, V / 32), block = 512: thread = (column v = 32 * blockIdx.y + tid / 16, k-slice ks = tid % 16: 8 of the 128 k).
// The recurrence is VALU-issue-bound (4 lane-ops per state element and token), so a value head is spread over 4 CUs and
// 16 lanes share a column: ~50 instructions per token and wave instead of 160 (4 lanes per column, one CU per head:
// 0.59 us per token).  The 16 partial sums of a column are one DPP row reduction.
__global__ __launch_bounds__(512) void gdn_scan_kernel(GdnArgs a) {
    constexpr int K = 128, V = 128, KS = 16, KP = K / KS, CB = 32, TB = 8;
    __shared__ __attribute__((aligned(16))) float qk[2][TB][2][K];      // double-buffered q^ / k^ of TB tokens
    __shared__ float bd[2][TB][2];
    const int h = blockIdx.x, tid = threadIdx.x, v = blockIdx.y * CB + (tid >> 4), ks = tid & 15;
    const int nk = a.NV / a.vpg;
    const int kh = a.chunked ? h % nk : h / a.vpg;
    float* Sg = a.state_pool + (((size_t)a.slot * a.gdn_layers + a.layer_idx) * a.NV + h) * K * V;
    float S[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) S[k] = Sg[(ks * KP + k) * V + v];
    // The next batch of TB tokens (q^ / k^ rows, beta / decay, this column's v values) is requested into REGISTERS before
    // the current batch is consumed and parked in LDS after it: the loads have a whole batch of compute to arrive.
    float rq[4], rbd = 0.f, vvn[TB], vv[TB];
    auto fetch = [&](int t0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 512 * i, tt = e >> 8, which = (e >> 7) & 1, kk = e & 127;
            const int t = min(t0 + tt, a.S - 1);
            rq[i] = (which ? a.pre_k : a.pre_q)[(size_t)t * a.key_dim + kh * K + kk];
        }
        if (tid < 2 * TB) rbd = a.pre_bd[((size_t)min(t0 + (tid >> 1), a.S - 1) * a.NV + h) * 2 + (tid & 1)];
#pragma unroll
        for (int i = 0; i < TB; ++i) vvn[i] = a.pre_v[((size_t)min(t0 + i, a.S - 1) * a.NV + h) * V + v];
    };
    auto park = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 512 * i, tt = e >> 8, which = (e >> 7) & 1, kk = e & 127;
            qk[buf][tt][which][kk] = rq[i];
        }
        if (tid < 2 * TB) bd[buf][tid >> 1][tid & 1] = rbd;
#pragma unroll
        for (int i = 0; i < TB; ++i) vv[i] = vvn[i];
    };
    fetch(0);
    park(0);
    __syncthreads();
    for (int t0 = 0; t0 < a.S; t0 += TB) {
        const int buf = (t0 / TB) & 1;
        fetch(min(t0 + TB, a.S - 1));                // unconditional (clamped): no load under a branch
        const int nt = min(TB, a.S - t0);
#pragma unroll
        for (int i = 0; i < TB; ++i) {
            if (i >= nt) break;
            const float beta = bd[buf][i][0], decay = bd[buf][i][1];
            const f32x4* kp = (const f32x4*)&qk[buf][i][1][ks * KP];
            const f32x4* qp = (const f32x4*)&qk[buf][i][0][ks * KP];
            const f32x4 ka = kp[0], kb = kp[1], qa = qp[0], qb = qp[1];
            const float kf[KP] = {ka[0], ka[1], ka[2], ka[3], kb[0], kb[1], kb[2], kb[3]};
            const float qf[KP] = {qa[0], qa[1], qa[2], qa[3], qb[0], qb[1], qb[2], qb[3]};
            float kv2[2] = {0.f, 0.f};
#pragma unroll
            for (int k = 0; k < KP; ++k) { S[k] *= decay; kv2[k & 1] += S[k] * kf[k]; }
            const float kv = row16_sum(kv2[0] + kv2[1]);          // the 16 k-slices of this column are one DPP row
            const float delta = (vv[i] - kv) * beta;
            float y2[2] = {0.f, 0.f};
#pragma unroll
            for (int k = 0; k < KP; ++k) { S[k] += kf[k] * delta; y2[k & 1] += S[k] * qf[k]; }
            const float y = row16_sum(y2[0] + y2[1]);
            if (ks == 0) a.out[(size_t)(t0 + i) * a.out_stride + h * V + v] = y;      // raw; gdn_post_kernel normalises
        }
        park(buf ^ 1);                               // the other buffer was consumed one iteration ago (barrier below)
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < KP; ++k) Sg[(ks * KP + k) * V + v] = S[k];
}

__global__ __launch_bounds__(128) void gdn_post_kernel(GdnArgs a) {
    constexpr int V = 128;
    __shared__ float red[2];
    const int t = blockIdx.x, h = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int conv_dim = 2 * a.key_dim + a.NV * V;
    float* o = a.out + (size_t)t * a.out_stride + h * V + tid;
    const float y = *o;
    const float sy = wave_sum(y * y);
    if (lane == 0) red[wave] = sy;
    __syncthreads();
    const float rms = 1.0f / sqrtf((red[0] + red[1]) / (float)V + a.eps);
    const float z = a.proj[(size_t)t * a.proj_stride + conv_dim + h * V + tid];
    *o = y * rms * a.gnorm_w[tid] * silu_f(z);
}

void launch_gdn(const GdnArgs& a, hipStream_t s) {
    // prompts: three passes (needs the scratch rows of ensure_prefill_buffers); decode steps and short tails: fused
    if (a.st == nullptr && a.pre_q != nullptr && a.S >= 16 && a.n_seq <= 1) {
        const int nk = a.NV / a.vpg;
        hipLaunchKernelGGL(gdn_pre_kernel, dim3(a.S, nk + a.NV), dim3(128), 0, s, a);
        hipLaunchKernelGGL(gdn_scan_kernel, dim3(a.NV, 4), dim3(512), 0, s, a);
        hipLaunchKernelGGL(gdn_post_kernel, dim3(a.S, a.NV), dim3(128), 0, s, a);
        return;
    }
    if (a.st != nullptr && a.S == 1 && a.gdn_scratch != nullptr && a.gdn_ticket != nullptr) {     // decode step: 4 workgroups per value head
        hipLaunchKernelGGL(gdn_decode_kernel, dim3(a.NV, 4, a.n_seq > 0 ? a.n_seq : 1), dim3(256), 0, s, a);
        return;
    }
    hipLaunchKernelGGL(gdn_kernel, dim3(a.NV, a.n_seq > 0 ? a.n_seq : 1), dim3(128), 0, s, a);
}

__global__ void bf16_to_f32_kernel(const uint16_t* __restrict__ src, float* __restrict__ dst, size_t n, float add) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = bf16_to_f32(src[i]) + add;
}
void launch_bf16_to_f32(const uint16_t* src, float* dst, size_t n, float add, hipStream_t s) {
    int blocks = (int)std::min<size_t>((n + 255) / 256, 1024);
    hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(blocks < 1 ? 1 : blocks), dim3(256), 0, s, src, dst, n, add);
}

}  // namespace cm
