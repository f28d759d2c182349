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
#include <cstdlib>

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
        if (a.ba_w == nullptr && tid == 0) {
            const float beta = 1.0f / (1.0f + expf(-pr[conv_dim + a.NV * V + h]));
            const float av = pr[conv_dim + a.NV * V + a.NV + h] + a.dt_bias[h];
            bd[0] = beta;
            bd[1] = expf(-expf(a.A_log[h]) * logf(1.0f + expf(av)));
        }
    }
    if (a.ba_w != nullptr) {
        // the head's b and a: two dot products of RMSNorm(x) * w with bf16 rows (the whole workgroup; the norm's scale is factored out)
        const float* xr = a.ba_x + (size_t)bq * a.ba_H;
        const uint16_t* wb = a.ba_w + (size_t)h * a.ba_H;
        const uint16_t* wa = a.ba_w + (size_t)(a.NV + h) * a.ba_H;
        float s2 = 0.f, sb = 0.f, sa = 0.f;
        for (int i0 = tid * 4; i0 < a.ba_H; i0 += 4096) {          // (four chunks' loads together: the sums keep their order)
            f32x4 xv4[4], nv4[4]; u32x2 pb4[4], pa4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = i0 + 1024 * j < a.ba_H ? i0 + 1024 * j : i0;
                xv4[j] = *(const f32x4*)(xr + i); nv4[j] = *(const f32x4*)(a.ba_nw + i);
                pb4[j] = *(const u32x2*)(wb + i); pa4[j] = *(const u32x2*)(wa + i);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (i0 + 1024 * j >= a.ba_H) break;
                const f32x4 xv = xv4[j], nv = nv4[j];
                const u32x2 pb = pb4[j], pa = pa4[j];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xn = xv[e] * nv[e];
                    const float fb = bf16_to_f32((uint16_t)((pb[e >> 1] >> (16 * (e & 1))) & 0xFFFFu));
                    const float fa = bf16_to_f32((uint16_t)((pa[e >> 1] >> (16 * (e & 1))) & 0xFFFFu));
                    s2 = fmaf(xv[e], xv[e], s2); sb = fmaf(xn, fb, sb); sa = fmaf(xn, fa, sa);
                }
            }
        }
        __shared__ float bared[3][4];
        const float t2 = wave_sum(s2), tb = wave_sum(sb), ta = wave_sum(sa);
        __syncthreads();                                           // (red[] of the q / k norms is written above: separate array, but order the phases)
        if (lane == 0) { bared[0][wave] = t2; bared[1][wave] = tb; bared[2][wave] = ta; }
        __syncthreads();
        if (tid == 0) {
            const float rr = 1.0f / sqrtf(((bared[0][0] + bared[0][1]) + (bared[0][2] + bared[0][3])) / (float)a.ba_H + a.eps);
            const float bv = rr * ((bared[1][0] + bared[1][1]) + (bared[1][2] + bared[1][3]));
            const float av = rr * ((bared[2][0] + bared[2][1]) + (bared[2][2] + bared[2][3])) + a.dt_bias[h];
            bd[0] = 1.0f / (1.0f + expf(-bv));
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
            if (a.pre_g != nullptr) a.pre_g[(size_t)t * a.NV + h] = g;
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


// ---------------------------------------------------------------------------------------------------------
// Chunk-parallel prompt scan (round 5): the gated delta rule in its WY / UT-transform form on the f32 matrix cores.
// The sequential scan above is bound by two dependent 16-lane reductions per token on 64 workgroups (240 us per layer and
// 1024 tokens on Qwen3.5-0.8B: 43 % of its prompt pass).  The reference names the chunked form as the prompt-path algorithm
// (ops/gdn/backend.rs:100-104); HF's torch_chunk_gated_delta_rule is the same algebra.  Per value head and chunk of C = 64
// tokens with state S0 [K x V] at the chunk's start (t, s index tokens of the chunk; G_t = g_1 + ... + g_t, g = log decay):
//     u_t = beta_t (v_t - e^{G_t} S0^T k_t - sum_{s<t} e^{G_t - G_s} (k_s . k_t) u_s)          (the delta rule, unrolled)
//  => (I + A) U = beta V - (beta e^G K) S0,   A[t][s] = beta_t e^{G_t - G_s} (k_t . k_s) for s < t
//  => U = U0 - W S0  with  W = (I + A)^-1 (beta e^G K),  U0 = (I + A)^-1 (beta V)              (independent of S0: all chunks in parallel)
//     Y = e^G (Q S0) + P U,   P[t][s] = e^{G_t - G_s} (q_t . k_s) for s <= t
//     S_C = e^{G_C} S0 + K^T (e^{G_C - G} U)
// gdn_chunk_prep_kernel (grid chunks x NV): A and P on v_mfma_f32_16x16x4_f32, (I + A)^-1 [.] by forward substitution (one
// thread per column of [beta e^G K | beta V], its 64 unknowns in registers: no dependency between threads) -> W, U0, P, G.
// gdn_chunk_scan_kernel (grid NV x V/16): the only sequential part -- three small matrix products per chunk over a 128 x 16
// slice of the state (columns of S are independent).  Everything stays f32 (f32 MFMA: products and sums in f32), logs of the
// decays are summed, never multiplied (a decay of e^-80 per token underflows a running product after two tokens).
// Same results as the sequential scan to f32 summation order (tests: prompts through both, logits < 1e-4).
// ---------------------------------------------------------------------------------------------------------
constexpr int CKP = 132;          // LDS row stride (floats) of a 128-wide row: 16-byte aligned, rows 4 banks apart
constexpr int CTP = 68;           // ... of a 64-wide row

// workgroup barrier for LDS hand-offs only: waits for this wave's LDS operations, NOT for its outstanding global loads
// (__syncthreads() carries a workgroup-scope fence that drains vmcnt as well -- it would retire the register prefetch of the
// next chunk at every barrier)
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ f32x4 mfma4(const f32x4& av, const f32x4& bv, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bv[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bv[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], bv[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], bv[3], c, 0, 0, 0);
    return c;
}

size_t gdn_chunk_prep_lds() { return (size_t)(2 * GDN_CK * CKP + GDN_CK * CTP + 2 * GDN_CK) * sizeof(float); }
size_t gdn_chunk_scan_lds() { return (size_t)(16 * CKP + 2 * 16 * CTP + GDN_CK * CKP + GDN_CK) * sizeof(float); }

__global__ __launch_bounds__(256) void gdn_chunk_prep_kernel(GdnArgs a) {
    constexpr int K = 128, V = 128, C = GDN_CK;
    extern __shared__ __attribute__((aligned(16))) float cl[];
    float (*Ks)[CKP] = (float (*)[CKP])cl;
    float (*Qs)[CKP] = (float (*)[CKP])(cl + C * CKP);
    float (*Am)[CTP] = (float (*)[CTP])(cl + 2 * C * CKP);
    float* lg = cl + 2 * C * CKP + C * CTP;
    float* bt = lg + C;
    const int c = blockIdx.x, h = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nk = a.NV / a.vpg, kh = a.chunked ? h % nk : h / a.vpg;
    const int t0 = c * C, nt = min(C, a.S - t0);
    float* ck = a.ck + ((size_t)c * a.NV + h) * GDN_CK_FLOATS;
    float* Wg = ck, *U0g = ck + C * 128, *Pg = ck + 2 * C * 128, *Gg = ck + 2 * C * 128 + C * C;
    if (wave == 0) {                                   // cumulative log decay of the chunk (inclusive), beta
        float g = lane < nt ? a.pre_g[(size_t)(t0 + lane) * a.NV + h] : 0.f;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const float o = __shfl_up(g, d); if (lane >= d) g += o; }
        lg[lane] = g;
        bt[lane] = lane < nt ? a.pre_bd[((size_t)(t0 + lane) * a.NV + h) * 2] : 0.f;
        Gg[lane] = g;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {                      // k^ and q^ rows of the chunk (rows past the prompt: zero)
        const int e = tid + 256 * i, row = e >> 5, c4 = (e & 31) << 2;
        f32x4 kv = {0.f, 0.f, 0.f, 0.f}, qv = {0.f, 0.f, 0.f, 0.f};
        if (row < nt) {
            kv = *(const f32x4*)(a.pre_k + (size_t)(t0 + row) * a.key_dim + kh * K + c4);
            qv = *(const f32x4*)(a.pre_q + (size_t)(t0 + row) * a.key_dim + kh * K + c4);
        }
        *(f32x4*)&Ks[row][c4] = kv;
        *(f32x4*)&Qs[row][c4] = qv;
    }
    __syncthreads();
    {   // A = tril(beta e^{G_t - G_s} K K^T, -1) -> LDS; P = tril(e^{G_t - G_s} Q K^T) -> scratch.  Wave w: token rows 16 w ..
        const int r = lane & 15, kq = lane >> 4;
        f32x4 accA[4], accP[4];
#pragma unroll
        for (int tc = 0; tc < 4; ++tc) { accA[tc] = (f32x4){0.f, 0.f, 0.f, 0.f}; accP[tc] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f32x4 ak = *(const f32x4*)&Ks[16 * wave + r][16 * j + 4 * kq];
            const f32x4 aq = *(const f32x4*)&Qs[16 * wave + r][16 * j + 4 * kq];
#pragma unroll
            for (int tc = 0; tc < 4; ++tc) {
                if (tc > wave) continue;               // (s > t everywhere in the tile)
                const f32x4 b = *(const f32x4*)&Ks[16 * tc + r][16 * j + 4 * kq];
                accA[tc] = mfma4(ak, b, accA[tc]);
                accP[tc] = mfma4(aq, b, accP[tc]);
            }
        }
#pragma unroll
        for (int tc = 0; tc < 4; ++tc)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int t = 16 * wave + 4 * kq + i, s2 = 16 * tc + r;
                const float e = s2 <= t ? expf(lg[t] - lg[s2]) : 0.f;
                Am[t][s2] = s2 < t ? bt[t] * e * accA[tc][i] : 0.f;
                Pg[t * C + s2] = e * accP[tc][i];
            }
    }
    lds_barrier();
    // ---- X = (I + A)^-1 [beta e^G K | beta V] by BLOCKED forward substitution (16-token blocks): the off-diagonal part
    // R_I = B_I - sum_{L<I} A_IL X_L on the matrix cores, the 16 x 16 diagonal solves one thread per column (120 FMAs each).
    // (All 64 rows per thread as scalar FMAs were LDS-bound: every thread reads every A[t][s] -- 512 broadcast reads per thread.)
    // X lives in LDS where k^ / q^ were ([64][264] = the same floats): every thread first forms its column of B in registers.
    {
        const bool isk = tid < K;
        float b[C];
        if (isk) {
#pragma unroll
            for (int t = 0; t < C; ++t) b[t] = bt[t] * expf(lg[t]) * Ks[t][tid];
        } else {
#pragma unroll
            for (int t = 0; t < C; ++t) b[t] = a.pre_v[((size_t)min(t0 + t, a.S - 1) * a.NV + h) * V + (tid - K)];
#pragma unroll
            for (int t = 0; t < C; ++t) b[t] = t < nt ? bt[t] * b[t] : 0.f;
        }
        lds_barrier();                                  // every read of Ks / Qs is done: the region becomes X
        constexpr int XP = 2 * CKP;                    // 264
        float (*X)[XP] = (float (*)[XP])cl;
#pragma unroll
        for (int t = 0; t < C; ++t) X[t][tid] = b[t];
        float* dst = (isk ? Wg : U0g) + (isk ? tid : tid - K);
        const int r = lane & 15, kq = lane >> 4;
#pragma unroll
        for (int I = 0; I < 4; ++I) {
            lds_barrier();
            if (I > 0) {                               // R_I: wave w owns column tiles 4 w .. 4 w + 3
                f32x4 acc[4];
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int L = 0; L < I; ++L) {
                    const f32x4 av = *(const f32x4*)&Am[16 * I + r][16 * L + 4 * kq];
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        f32x4 bv;
#pragma unroll
                        for (int i = 0; i < 4; ++i) bv[i] = X[16 * L + 4 * kq + i][16 * (4 * wave + n) + r];
                        acc[n] = mfma4(av, bv, acc[n]);
                    }
                }
#pragma unroll
                for (int n = 0; n < 4; ++n)
#pragma unroll
                    for (int i = 0; i < 4; ++i) X[16 * I + 4 * kq + i][16 * (4 * wave + n) + r] -= acc[n][i];
                lds_barrier();
            }
            float x[16];                               // diagonal block: x_t = r_t - sum_{s<t, same block} A[t][s] x_s
#pragma unroll
            for (int t = 0; t < 16; ++t) x[t] = X[16 * I + t][tid];
#pragma unroll
            for (int t = 1; t < 16; ++t) {
                float p0 = 0.f, p1 = 0.f;
#pragma unroll
                for (int s4 = 0; s4 < t; s4 += 4) {
                    const f32x4 av = *(const f32x4*)&Am[16 * I + t][16 * I + s4];
                    p0 += av[0] * x[s4];
                    if (s4 + 1 < t) p1 += av[1] * x[s4 + 1];
                    if (s4 + 2 < t) p0 += av[2] * x[s4 + 2];
                    if (s4 + 3 < t) p1 += av[3] * x[s4 + 3];
                }
                x[t] -= p0 + p1;
            }
#pragma unroll
            for (int t = 0; t < 16; ++t) { X[16 * I + t][tid] = x[t]; dst[(16 * I + t) * 128] = x[t]; }
        }
    }
}

__global__ __launch_bounds__(256) void gdn_chunk_scan_kernel(GdnArgs a) {
    constexpr int K = 128, V = 128, C = GDN_CK, VB = 16;
    extern __shared__ __attribute__((aligned(16))) float cl[];
    float (*St)[CKP] = (float (*)[CKP])cl;                              // state slice, transposed: [column][k]
    float (*Ut)[CTP] = (float (*)[CTP])(cl + VB * CKP);                 // U of the chunk, transposed: [column][t]
    float (*Vt)[CTP] = (float (*)[CTP])(cl + VB * CKP + VB * CTP);      // e^{G_C - G_t} U
    float (*Ks)[CKP] = (float (*)[CKP])(cl + VB * CKP + 2 * VB * CTP);  // k^ rows of the chunk
    float* lgs = cl + VB * CKP + 2 * VB * CTP + C * CKP;
    const int h = blockIdx.x, cb = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, kq = lane >> 4;
    const int nk = a.NV / a.vpg, kh = a.chunked ? h % nk : h / a.vpg;
    float* Sg = a.state_pool + (((size_t)a.slot * a.gdn_layers + a.layer_idx) * a.NV + h) * K * V;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int e = tid + 256 * i, k = e >> 4, col = e & 15;
        St[col][k] = Sg[k * V + VB * cb + col];
    }
    const int nchunks = (a.S + C - 1) / C;
    // One wave per SIMD: nothing hides a load but the wave's own work, so the operands of chunk c + 1 (this wave's 16 rows of W,
    // q^, P, its U0 values, its share of the k^ rows, the cumulative log decay) are requested into registers while chunk c is
    // multiplied -- unconditionally (rows / chunks past the end are clamped and never consumed: DESIGN 3.13).
    // Register sets: W / q^ rows (needed FIRST in a chunk) are requested TWO chunks ahead into two alternating sets, the rest
    // (P rows, U0 values, k^ rows, cumulative log decay: needed later in the chunk) one chunk ahead.
    f32x4 awA[8], aqA[8], awB[8], aqB[8], ap[4], kreg[8];
    float u0[4], glane = 0.f;
    auto fetch_wq = [&](int c, f32x4 (&aw)[8], f32x4 (&aq)[8]) __attribute__((always_inline)) {
        const int cc = min(c, nchunks - 1), t0 = cc * C;
        const float* ck = a.ck + ((size_t)cc * a.NV + h) * GDN_CK_FLOATS;
        const int qrow = min(t0 + 16 * wave + r, a.S - 1);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            aw[j] = *(const f32x4*)(ck + (16 * wave + r) * 128 + 16 * j + 4 * kq);
            aq[j] = *(const f32x4*)(a.pre_q + (size_t)qrow * a.key_dim + kh * K + 16 * j + 4 * kq);
        }
    };
    auto fetch_pu = [&](int c) __attribute__((always_inline)) {         // P rows + U0 values of chunk c (issued once the previous ones are consumed)
        const int cc = min(c, nchunks - 1);
        const float* ck = a.ck + ((size_t)cc * a.NV + h) * GDN_CK_FLOATS;
#pragma unroll
        for (int j = 0; j < 4; ++j) ap[j] = *(const f32x4*)(ck + 2 * C * 128 + (16 * wave + r) * C + 16 * j + 4 * kq);
#pragma unroll
        for (int i = 0; i < 4; ++i) u0[i] = ck[C * 128 + (16 * wave + 4 * kq + i) * 128 + VB * cb + r];
    };
    auto fetch_k = [&](int c) __attribute__((always_inline)) {          // k^ rows + cumulative log decay of chunk c: parked in LDS at the END of
        const int cc = min(c, nchunks - 1), t0 = cc * C;                 //   chunk c - 1, so they are requested at its very beginning
        const float* ck = a.ck + ((size_t)cc * a.NV + h) * GDN_CK_FLOATS;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + 256 * i, row = min(t0 + (e >> 5), a.S - 1), c4 = (e & 31) << 2;
            kreg[i] = *(const f32x4*)(a.pre_k + (size_t)row * a.key_dim + kh * K + c4);
        }
        glane = ck[2 * C * 128 + C * C + lane];
    };
    auto park_k = [&](int c) __attribute__((always_inline)) {          // k^ rows + cumulative log decay of chunk c -> LDS
        const int nt = min(C, a.S - c * C);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + 256 * i, row = e >> 5, c4 = (e & 31) << 2;
            *(f32x4*)&Ks[row][c4] = row < nt ? kreg[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (wave == 0) lgs[lane] = glane;
    };
    auto chunk = [&](int c, f32x4 (&aw)[8], f32x4 (&aq)[8]) __attribute__((always_inline)) {
        const int t0 = c * C, nt = min(C, a.S - t0);
        fetch_k(c + 1);
        // ---- U = U0 - W S0 and Q S0 (same B operand): wave w owns token rows 16 w .. 16 w + 15 ----
        f32x4 accU = {0.f, 0.f, 0.f, 0.f}, accQ = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f32x4 bs = *(const f32x4*)&St[r][16 * j + 4 * kq];
            accU = mfma4(aw[j], bs, accU);
            accQ = mfma4(aq[j], bs, accQ);
        }
        fetch_wq(c + 2, aw, aq);                       // this set is free again: the chunk after next
        const float gC = lgs[C - 1];
        f32x4 u, ud, pc[4];
        float eg[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t = 16 * wave + 4 * kq + i;
            const float gt = lgs[t];
            u[i] = u0[i] - accU[i];
            ud[i] = u[i] * expf(gC - gt);
            eg[i] = expf(gt);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) pc[j] = ap[j];
        *(f32x4*)&Ut[r][16 * wave + 4 * kq] = u;
        *(f32x4*)&Vt[r][16 * wave + 4 * kq] = ud;
        fetch_pu(c + 1);                               // (ap / u0 of this chunk are consumed or copied)
        lds_barrier();
        // ---- Y = e^G (Q S0) + P U ----
        f32x4 accP = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 bu = *(const f32x4*)&Ut[r][16 * j + 4 * kq];
            accP = mfma4(pc[j], bu, accP);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t = 16 * wave + 4 * kq + i;
            if (t < nt) a.out[(size_t)(t0 + t) * a.out_stride + h * V + VB * cb + r] = eg[i] * accQ[i] + accP[i];      // raw; gdn_post_kernel normalises
        }
        // ---- S_C = e^{G_C} S0 + K^T (e^{G_C - G} U): wave w owns state rows k = 32 w .. 32 w + 31 ----
        f32x4 sn[2];
        const float eC = expf(gC);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            f32x4 accS = {0.f, 0.f, 0.f, 0.f};
            const int kr = 32 * wave + 16 * mt;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 bv = *(const f32x4*)&Vt[r][16 * j + 4 * kq];
                f32x4 ak;
#pragma unroll
                for (int i = 0; i < 4; ++i) ak[i] = Ks[16 * j + 4 * kq + i][kr + r];
                accS = mfma4(ak, bv, accS);
            }
            const f32x4 old = *(const f32x4*)&St[r][kr + 4 * kq];
#pragma unroll
            for (int i = 0; i < 4; ++i) sn[mt][i] = eC * old[i] + accS[i];
        }
        lds_barrier();                               // every read of St / Ut / Vt / Ks / lgs of this chunk is done
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) *(f32x4*)&St[r][32 * wave + 16 * mt + 4 * kq] = sn[mt];
        park_k(c + 1);
        lds_barrier();
    };
    fetch_wq(0, awA, aqA);
    fetch_pu(0);
    fetch_k(0);
    fetch_wq(1, awB, aqB);
    park_k(0);
    lds_barrier();
    for (int c = 0; c < nchunks; c += 2) {
        chunk(c, awA, aqA);
        if (c + 1 < nchunks) chunk(c + 1, awB, aqB);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int e = tid + 256 * i, k = e >> 4, col = e & 15;
        Sg[k * V + VB * cb + col] = St[col][k];
    }
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
        // prompts of a chunk or more: the chunk-parallel scan on the f32 matrix cores (CM_GDN_CHUNKED=0: the sequential scan, A/B)
        static const int ck_env = getenv("CM_GDN_CHUNKED") ? atoi(getenv("CM_GDN_CHUNKED")) : 1;
        if (a.ck != nullptr && a.pre_g != nullptr && ck_env != 0 && a.S >= GDN_CK) {
            static DevOnce attr;
            attr.run([&] {
                (void)hipFuncSetAttribute((const void*)gdn_chunk_prep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gdn_chunk_prep_lds());
                (void)hipFuncSetAttribute((const void*)gdn_chunk_scan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gdn_chunk_scan_lds());
            });
            hipLaunchKernelGGL(gdn_chunk_prep_kernel, dim3((a.S + GDN_CK - 1) / GDN_CK, a.NV), dim3(256), gdn_chunk_prep_lds(), s, a);
            hipLaunchKernelGGL(gdn_chunk_scan_kernel, dim3(a.NV, 128 / 16), dim3(256), gdn_chunk_scan_lds(), s, a);
        } else
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
