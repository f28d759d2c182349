// Quantised-weight decode kernels: GEMV over ggml Q8_0 / Q4_K / Q6_K weights, the quantised
// embedding-row gather, and the in-situ Q8_0 quantiser.
//
// Replaces LinearLayer::Quantized(QMatMul) (crane-core/src/ops/linear.rs:18-51: "dequantizes weights to F32 and
// requires F32 input") and EmbeddingLayer::Quantized (modules/embedding.rs:31-98): y[n] = sum_k dequant(W)[n,k] * x[k]
// with f32 activations and f32 accumulation -- the weights are decoded in registers on the way from HBM and never
// materialised.  Block arithmetic follows ggml's dequantize_row_{q8_0,q4_K,q6_K} (restated in oracle/gguf_oracle.py).
//
// HBM layouts (repacked once at load from the GGUF byte stream; pure permutations, no re-quantisation):
//   Q8_0 : codes i8 [N][K]                       | d f16 [N][K/32]
//   Q4_K : qs u8 [N][K/2] (128 B per 256-block, native nibble order) | hdr [N][K/256][16 B] = {f16 d, f16 dmin, u8 scales[12]}
//   Q6_K : ql u8 [N][K/2] | qh u8 [N][K/4] | sc i8 [N][K/16] | d f16 [N][K/256]
// so every stream a wave reads is contiguous and 16-byte aligned (the GGUF blocks are 34 / 144 / 210 bytes).
//
// Work split: a wave owns R = 2 rows and sweeps K in chunks of CK elements (1024 for Q8_0: 16 codes = 16 B per
// lane; 2048 for the K-quants: 8 lanes per 256-block, 32 weights per lane).  x lives in LDS as f32, permuted per
// format so that the NJ float4 a lane needs per chunk are at [chunk][j][lane] (conflict-free ds_read_b128).
// Same fused prologue (RMSNorm) and epilogues (store / residual add / SiLU*mul / arg-max) as the bf16 GEMV.
#include <stdexcept>

#include "dev_common.h"
#include "kernels.h"

namespace cm {

__device__ __forceinline__ float f16_bits_to_f32(uint32_t h) {
    _Float16 v;
    const uint16_t b = (uint16_t)h;
    __builtin_memcpy(&v, &b, 2);
    return (float)v;
}
__device__ __forceinline__ float sb(uint32_t w, int n) { return (float)(int)(signed char)((w >> (8 * n)) & 0xFFu); }
__device__ __forceinline__ float ub(uint32_t w, int n) { return (float)((w >> (8 * n)) & 0xFFu); }

template <int FMT> struct QF;
template <> struct QF<QFMT_Q8_0> { static constexpr int CK = 1024, NJ = 4, SH = 8; };
template <> struct QF<QFMT_Q4_K> { static constexpr int CK = 2048, NJ = 8, SH = 9; };
template <> struct QF<QFMT_Q6_K> { static constexpr int CK = 2048, NJ = 8, SH = 9; };

// LDS float4 slot of x[4*k4 .. 4*k4+3]
template <int FMT>
__device__ __forceinline__ int x_slot(int k4) {
    if (FMT == QFMT_Q8_0) {
        const int c = k4 >> 8, r = k4 & 255;
        return c * 256 + (r & 3) * 64 + (r >> 2);
    } else if (FMT == QFMT_Q4_K) {
        const int c = k4 >> 9, r = k4 & 511, b = r >> 6, e4 = r & 63, p = e4 >> 4, f4 = e4 & 15;
        const int hi = f4 >> 3, g4 = f4 & 7, half = g4 >> 2, q = g4 & 3;
        return c * 512 + (hi * 4 + q) * 64 + (b * 8 + p * 2 + half);
    } else {
        const int c = k4 >> 9, r = k4 & 511, b = r >> 6, e4 = r & 63, n = e4 >> 5, f4 = e4 & 31;
        const int t = f4 >> 3, g4 = f4 & 7, j = g4 >> 1, q = g4 & 1;
        return c * 512 + (t * 2 + q) * 64 + (b * 8 + n * 4 + j);
    }
}

// The bytes one lane needs for one (row, chunk), exactly as loaded: nothing is converted or re-packed at load time, so
// the loads carry no ALU dependency and can stay in flight while the previous (row, chunk) is consumed.
struct QRow {
    u32x4 a;                  // Q8_0: 16 codes | Q4_K: 16 B of qs | Q6_K: {ql[l], ql[l+32]} 8 B each
    u32x4 b;                  // Q4_K: block header | Q6_K: {qh 8 B, the 8 scale bytes of this half-block}
    uint32_t dh;              // Q8_0 / Q6_K: block scale d, raw f16 bits
};
__device__ __forceinline__ float q_d(const QRow& r) { return f16_bits_to_f32(r.dh); }
// Q6_K: scale of run t (weights 128n + 32t + 8j ..): byte (j >> 1) + 2t of the half-block's 8 scale bytes
__device__ __forceinline__ int q6_sc(const QRow& r, int t, int lane) {
    const int bi = ((lane >> 1) & 1) + 2 * (t & 1);
    return (int)(signed char)((r.b[2 + (t >> 1)] >> (8 * bi)) & 0xFFu);
}

template <int FMT>
__device__ __forceinline__ QRow q_load(const QWeight& w, int row, int c, int lane) {
    QRow r;
    r.dh = 0;
    const size_t K = (size_t)w.K;
    if (FMT == QFMT_Q8_0) {
        const size_t k = (size_t)c * 1024 + (size_t)lane * 16;
        r.a = ld_nt16(w.p0 + (size_t)row * K + k);
        r.b = (u32x4){0, 0, 0, 0};
        r.dh = *(const uint16_t*)(w.p1 + ((size_t)row * (K >> 5) + (k >> 5)) * 2);
    } else if (FMT == QFMT_Q4_K) {
        const size_t blk = (size_t)row * (K >> 8) + (size_t)c * 8 + (lane >> 3);
        const int h = lane & 7;
        r.a = ld_nt16(w.p0 + blk * 128 + (h >> 1) * 32 + (h & 1) * 16);
        r.b = ld16(w.p1 + blk * 16);
    } else {
        const size_t blk = (size_t)row * (K >> 8) + (size_t)c * 8 + (lane >> 3);
        const int n = (lane >> 2) & 1, j = lane & 3;
        const u32x2 q0 = *(const u32x2*)(w.p0 + blk * 128 + n * 64 + j * 8);
        const u32x2 q1 = *(const u32x2*)(w.p0 + blk * 128 + n * 64 + 32 + j * 8);
        const u32x2 qh = *(const u32x2*)(w.p1 + blk * 64 + n * 32 + j * 8);
        r.a = (u32x4){q0[0], q0[1], q1[0], q1[1]};
        const u32x2 sc = *(const u32x2*)(w.p2 + blk * 16 + n * 8);
        r.b = (u32x4){qh[0], qh[1], sc[0], sc[1]};
        r.dh = *(const uint16_t*)(w.p3 + blk * 2);
    }
    return r;
}

// sum over this lane's weights of dequant(w) * x ; xs = the lane's NJ float4, sx = per-run sums of x
template <int FMT>
__device__ __forceinline__ float q_dot(const QRow& r, const f32x4* xv, const float* sx, int lane) {
    if (FMT == QFMT_Q8_0) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            s += sb(r.a[j], 0) * xv[j][0] + sb(r.a[j], 1) * xv[j][1] + sb(r.a[j], 2) * xv[j][2] + sb(r.a[j], 3) * xv[j][3];
        return q_d(r) * s;
    } else if (FMT == QFMT_Q4_K) {
        const float d = f16_bits_to_f32(r.b[0] & 0xFFFFu), dmin = f16_bits_to_f32(r.b[0] >> 16);
        // scales[12] = bytes 4..15 of the header; get_scale_min_k4 for sub-blocks 2p (low nibbles), 2p+1 (high)
        const int p = (lane & 7) >> 1;
        auto sbyte = [&](int i) -> uint32_t { const int bi = 4 + i; return (r.b[bi >> 2] >> (8 * (bi & 3))) & 0xFFu; };
        auto scale_min = [&](int j, float& sc, float& mn) {
            if (j < 4) { sc = (float)(sbyte(j) & 63u); mn = (float)(sbyte(j + 4) & 63u); }
            else {
                sc = (float)((sbyte(j + 4) & 0xFu) | ((sbyte(j - 4) >> 6) << 4));
                mn = (float)((sbyte(j + 4) >> 4) | ((sbyte(j) >> 6) << 4));
            }
        };
        float sc0, m0, sc1, m1;
        scale_min(2 * p, sc0, m0);
        scale_min(2 * p + 1, sc1, m1);
        float lo = 0.f, hi = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t wl = r.a[j] & 0x0F0F0F0Fu, wh = (r.a[j] >> 4) & 0x0F0F0F0Fu;
            lo += ub(wl, 0) * xv[j][0] + ub(wl, 1) * xv[j][1] + ub(wl, 2) * xv[j][2] + ub(wl, 3) * xv[j][3];
            hi += ub(wh, 0) * xv[4 + j][0] + ub(wh, 1) * xv[4 + j][1] + ub(wh, 2) * xv[4 + j][2] + ub(wh, 3) * xv[4 + j][3];
        }
        return (d * sc0) * lo - (dmin * m0) * sx[0] + (d * sc1) * hi - (dmin * m1) * sx[1];
    } else {
        const float d = q_d(r);
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            // run t: weights 128n + 32t + 8j + i ; low 4 bits from ql (t<2: low nibble, t>=2: high nibble of ql[l] / ql[l+32])
            const int qsel = (t & 1) * 2;                  // t odd -> ql[l + 32] words (a[2], a[3])
            const int hshift = 2 * t;
            float s = 0.f;
#pragma unroll
            for (int wi = 0; wi < 2; ++wi) {
                const uint32_t qlw = r.a[qsel + wi], qhw = r.b[wi];
                const uint32_t lo4 = (t < 2 ? qlw : (qlw >> 4)) & 0x0F0F0F0Fu;
                const uint32_t hi2 = ((qhw >> hshift) & 0x03030303u) << 4;
                const uint32_t q = lo4 | hi2;
                const f32x4 x = xv[t * 2 + wi];
                s += ub(q, 0) * x[0] + ub(q, 1) * x[1] + ub(q, 2) * x[2] + ub(q, 3) * x[3];
            }
            const float sc = (float)q6_sc(r, t, lane);
            acc += (d * sc) * (s - 32.0f * sx[t]);
        }
        return acc;
    }
}

template <int FMT, int PRO, int EPI>
__global__ __launch_bounds__(256, 4) void gemvq_kernel(GemvQArgs a) {
    using F = QF<FMT>;
    constexpr int R = 2, NJ = F::NJ, CK = F::CK;
    extern __shared__ __attribute__((aligned(16))) float xs[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = a.w.K, N = a.w.N;
    const int nch = (K + CK - 1) / CK;
    float* red = xs + (size_t)nch * CK;

    // ---- stage x into LDS (format permutation), fused RMSNorm statistics ----
    float ss = 0.f;
    const int n4 = K >> 2;
    for (int k4 = tid; k4 < n4; k4 += 256) {
        f32x4 v = *(const f32x4*)(a.x + (k4 << 2));
        if (PRO == PRO_RMSNORM) {
            ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
            const f32x4 w = *(const f32x4*)(a.nw + (k4 << 2));
            v[0] *= w[0]; v[1] *= w[1]; v[2] *= w[2]; v[3] *= w[3];
        }
        ((f32x4*)xs)[x_slot<FMT>(k4)] = v;
    }
    float scale = 1.f;
    if (PRO == PRO_RMSNORM) {
        ss = wave_sum(ss);
        if (lane == 0) red[wave] = ss;
    }
    __syncthreads();
    if (PRO == PRO_RMSNORM) scale = 1.0f / sqrtf(((red[0] + red[1]) + (red[2] + red[3])) / (float)K + a.eps);

    // lanes past the end of K in the last chunk stay idle (K % 32 == 0 for Q8_0, K % 256 == 0 for the K-quants)
    const int lane_k = (FMT == QFMT_Q8_0) ? lane * 16 : (lane >> 3) * 256;
    float best = -INFINITY; int besti = 0x7FFFFFFF;
    const int G = (N + R - 1) / R;
    for (int g = blockIdx.x * 4 + wave; g < G; g += gridDim.x * 4) {
        const int r0 = g * R;
        float acc[R];
#pragma unroll
        for (int i = 0; i < R; ++i) acc[i] = 0.f;
        for (int c = 0; c < nch; ++c) {
            if (c * CK + lane_k >= K) continue;
            QRow q[R];
#pragma unroll
            for (int i = 0; i < R; ++i) q[i] = q_load<FMT>(a.w, (r0 + i < N) ? r0 + i : N - 1, c, lane);
            f32x4 xv[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) xv[j] = ((const f32x4*)xs)[c * (CK / 4) + j * 64 + lane];
            float sx[4] = {0.f, 0.f, 0.f, 0.f};
            if (FMT == QFMT_Q4_K) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    sx[0] += (xv[j][0] + xv[j][1]) + (xv[j][2] + xv[j][3]);
                    sx[1] += (xv[4 + j][0] + xv[4 + j][1]) + (xv[4 + j][2] + xv[4 + j][3]);
                }
            } else if (FMT == QFMT_Q6_K) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    sx[t] = ((xv[2 * t][0] + xv[2 * t][1]) + (xv[2 * t][2] + xv[2 * t][3])) +
                            ((xv[2 * t + 1][0] + xv[2 * t + 1][1]) + (xv[2 * t + 1][2] + xv[2 * t + 1][3]));
            }
#pragma unroll
            for (int i = 0; i < R; ++i) acc[i] += q_dot<FMT>(q[i], xv, sx, lane);
        }
        float mine = 0.f, mine_up = 0.f;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            acc[i] = wave_sum(acc[i]) * scale;
            if (EPI != EPI_SILUMUL && lane == i) mine = acc[i];
            if (EPI == EPI_SILUMUL && (i & 1) && lane == (i >> 1)) { mine = acc[i - 1]; mine_up = acc[i]; }
        }
        if (EPI == EPI_STORE) {
            if (lane < R && r0 + lane < N) a.y[r0 + lane] = mine;
        } else if (EPI == EPI_RESADD) {
            if (lane < R && r0 + lane < N) a.y[r0 + lane] = a.res[r0 + lane] + mine;
        } else if (EPI == EPI_SILUMUL) {
            if (lane < R / 2 && r0 + 2 * lane + 1 < N) a.y[(r0 >> 1) + lane] = (mine / (1.0f + expf(-mine))) * mine_up;
        } else if (EPI == EPI_ARGMAX) {
            if (lane < R && r0 + lane < N) a.y[r0 + lane] = mine;
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const int ix = r0 + i + a.idx_base;
                if (r0 + i < N && (acc[i] > best || (acc[i] == best && ix < besti))) { best = acc[i]; besti = ix; }
            }
        }
    }
    if (EPI == EPI_ARGMAX) {
        __syncthreads();
        int* redi = (int*)(red + 4);
        if (lane == 0) { red[wave] = best; redi[wave] = besti; }
        __syncthreads();
        if (tid == 0) {
            float bb = red[0]; int bbi = redi[0];
            for (int w = 1; w < 4; ++w)
                if (red[w] > bb || (red[w] == bb && redi[w] < bbi)) { bb = red[w]; bbi = redi[w]; }
            a.pmax[blockIdx.x] = bb; a.pidx[blockIdx.x] = bbi;
        }
    }
}

QWeight QWeight::rows(int row0, int n) const {
    QWeight v = *this;
    v.N = n;
    const size_t r = (size_t)row0, k = (size_t)K;
    if (fmt == QFMT_Q8_0) { v.p0 = p0 + r * k; v.p1 = p1 + r * (k >> 5) * 2; }
    else if (fmt == QFMT_Q4_K) { v.p0 = p0 + r * (k >> 1); v.p1 = p1 + r * (k >> 8) * 16; }
    else if (fmt == QFMT_Q6_K) { v.p0 = p0 + r * (k >> 1); v.p1 = p1 + r * (k >> 2); v.p2 = p2 + r * (k >> 4); v.p3 = p3 + r * (k >> 8) * 2; }
    return v;
}
uint64_t QWeight::bytes() const {
    const uint64_t n = (uint64_t)N * (uint64_t)K;
    if (fmt == QFMT_Q8_0) return n + n / 32 * 2;
    if (fmt == QFMT_Q4_K) return n / 2 + n / 256 * 16;
    if (fmt == QFMT_Q6_K) return n / 2 + n / 4 + n / 16 + n / 256 * 2;
    return 0;
}

__global__ void silu_mul_kernel(const float* __restrict__ g, const float* __restrict__ u, float* __restrict__ o, int n,
                                int in_stride, int out_stride) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    g += (size_t)blockIdx.y * in_stride; u += (size_t)blockIdx.y * in_stride; o += (size_t)blockIdx.y * out_stride;
    if (i < n) { const float v = g[i]; o[i] = (v / (1.0f + expf(-v))) * u[i]; }
}
void launch_silu_mul(const float* gate, const float* up, float* out, int n, hipStream_t s, int n_seq, int in_stride, int out_stride) {
    hipLaunchKernelGGL(silu_mul_kernel, dim3((n + 255) / 256, n_seq), dim3(256), 0, s, gate, up, out, n, in_stride, out_stride);
}


// ---------------------------------------------------------------------------------------------------------
// Integer-dot GEMV: the activation row is quantised to the weight type's VecDotType in the prologue (Q8_0 blocks
// of 32 for Q8_0 weights, Q8_K blocks of 256 for the K-quants -- ggml quantize_row_q8_0 / quantize_row_q8_K as
// candle's k_quants `matmul` does before `vec_dot`), products are summed as INTEGERS inside a block with
// v_dot4_i32_i8 (4 MACs per lane-op instead of cvt + fma per weight) and scaled once per block:
//   Q8_0 . Q8_0 : sumf += (d_w * d_x) * sum(q_w * q_x)                                   (ggml_vec_dot_q8_0_q8_0)
//   Q4_K . Q8_K : sumf += (d_x * d) * sum_j sc_j * sum(q4 * q8) - (d_x * dmin) * sum_j m_j * bsum_j   (_q4_K_q8_K)
//   Q6_K . Q8_K : sumf += (d_x * d) * sum_j sc_j * sum((q6 - 32) * q8)                    (_q6_K_q8_K)
// x codes live in LDS in natural order (1 byte per element + per-block scales + per-8 code sums).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float f16_round(float v) {
    const _Float16 h = (_Float16)v;
    return (float)h;
}

#ifndef CM_Q8_R
#define CM_Q8_R 2
#endif
template <int FMT, int PRO, int EPI>
__global__ __launch_bounds__(256, 4) void gemvq_i8_kernel(GemvQArgs a) {
    using F = QF<FMT>;
    constexpr int R = FMT == QFMT_Q8_0 ? CM_Q8_R : 2, CK = F::CK;
    constexpr bool KQ = FMT != QFMT_Q8_0;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = a.w.K, N = a.w.N;
    const int nch = (K + CK - 1) / CK;
    const int Kpad = nch * CK;
    signed char* xq = (signed char*)lds_raw;                       // [Kpad] int8 codes
    float* xd = (float*)(lds_raw + Kpad);                          // block scales: [Kpad/32] (Q8_0) | [Kpad/256] (Q8_K)
    int* xs8 = (int*)(xd + (KQ ? Kpad / 256 : Kpad / 32));         // K-quants: sums of 8 consecutive codes [Kpad/8]
    float* red = (float*)(xs8 + (KQ ? Kpad / 8 : 0));

    // Weight loads are UNCONDITIONAL and carry no ALU on the loaded bytes: a lane (or a whole step) past the end of K
    // or N re-reads an in-bounds (row, chunk) whose product is zero (x codes, scales and code sums are zero past K) or
    // never consumed.  A load under a branch makes the compiler wait with vmcnt(0) at the first use -- which also waits
    // for the NEXT (row group, chunk) just requested and serialises the double buffer.
    const int lane_k = (FMT == QFMT_Q8_0) ? lane * 16 : (lane >> 3) * 256;
    const int G = (N + R - 1) / R;
    const int gstride = gridDim.x * 4;
    auto load_rows = [&](QRow (&dst)[R], int g, int c) {
        const bool inr = c * CK + lane_k < K;
        const int ce = inr ? c : 0, le = inr ? lane : (KQ ? (lane & 7) : 0);
        const int r0 = min(g, G - 1) * R;
#pragma unroll
        for (int i = 0; i < R; ++i) dst[i] = q_load<FMT>(a.w, min(r0 + i, N - 1), ce, le);
    };
    QRow qa[R], qb[R];
    const int gfirst = blockIdx.x * 4 + wave;
    load_rows(qa, gfirst, 0);              // requested before the activation row is quantised (independent of x)

    const int n4 = K >> 2;
    float rr = 1.f;
    // (round 6) the row is read ONCE: the values of the sum-of-squares pass stay in registers for the quantiser pass (rows of <= 8192
    // elements; its second read was one more dependent L2 round trip in front of the first dot product of every workgroup)
    constexpr int XKEEP = 8;
    const bool xkeep = PRO == PRO_RMSNORM && !KQ && n4 <= XKEEP * 256;
    f32x4 xr[XKEEP], wr[XKEEP];                                    // (and the norm weights: the second pass issues no global load)
    if (PRO == PRO_RMSNORM) {                                      // 1/rms first: the reference quantises the NORMALISED row
        float ss = 0.f;
        if (xkeep) {
#pragma unroll
            for (int i = 0; i < XKEEP; ++i) {
                const int k4 = tid + 256 * i;
                xr[i] = *(const f32x4*)(a.x + ((k4 < n4 ? k4 : tid) << 2));      // (unconditional loads; past the row: never used)
                wr[i] = *(const f32x4*)(a.nw + ((k4 < n4 ? k4 : tid) << 2));
            }
#pragma unroll
            for (int i = 0; i < XKEEP; ++i)
                if (tid + 256 * i < n4) { const f32x4 v = xr[i]; ss = fmaf(v[3], v[3], fmaf(v[2], v[2], fmaf(v[1], v[1], fmaf(v[0], v[0], ss)))); }
        } else {
            for (int k0 = tid; k0 < n4; k0 += 1024) {              // (four chunks' loads together; the sum keeps its order)
                f32x4 xb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) xb[i] = *(const f32x4*)(a.x + ((k0 + 256 * i < n4 ? k0 + 256 * i : k0) << 2));
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (k0 + 256 * i < n4) { const f32x4 v = xb[i]; ss = fmaf(v[3], v[3], fmaf(v[2], v[2], fmaf(v[1], v[1], fmaf(v[0], v[0], ss)))); }   // explicit: gemvqb must match bit for bit
            }
        }
        ss = wave_sum(ss);
        if (lane == 0) red[wave] = ss;
        __syncthreads();
        rr = 1.0f / sqrtf(((red[0] + red[1]) + (red[2] + red[3])) / (float)K + a.eps);
    }
    auto xnormw = [&](f32x4 v, const f32x4& w) -> f32x4 {
        v[0] = __fmul_rn(__fmul_rn(v[0], rr), w[0]); v[1] = __fmul_rn(__fmul_rn(v[1], rr), w[1]);
        v[2] = __fmul_rn(__fmul_rn(v[2], rr), w[2]); v[3] = __fmul_rn(__fmul_rn(v[3], rr), w[3]);
        return v;
    };
    if (K < Kpad) {                                                // lanes past K multiply zeros (see load_rows)
        for (int e = (K >> 2) + tid; e < (Kpad >> 2); e += 256) ((uint32_t*)xq)[e] = 0;
        const int sb0 = KQ ? K >> 8 : K >> 5, sb1 = KQ ? Kpad >> 8 : Kpad >> 5;
        for (int e = sb0 + tid; e < sb1; e += 256) xd[e] = 0.f;
        if (KQ) for (int e = (K >> 3) + tid; e < (Kpad >> 3); e += 256) xs8[e] = 0;
    }
    if (!KQ) {
        // quantize_row_q8_0: 32-element blocks = 8 consecutive lanes
        auto quant4 = [&](int k4, const f32x4& v) __attribute__((always_inline)) {
            float am = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
            am = fmaxf(am, __shfl_xor(am, 1)); am = fmaxf(am, __shfl_xor(am, 2)); am = fmaxf(am, __shfl_xor(am, 4));
            const float d = am / 127.0f;
            const float id = d != 0.f ? 1.0f / d : 0.f;
            uint32_t pk = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk |= ((uint32_t)(int)roundf(v[e] * id) & 0xFFu) << (8 * e);
            ((uint32_t*)xq)[k4] = pk;
            if ((tid & 7) == 0) xd[k4 >> 3] = f16_round(d);
        };
        if (xkeep) {
#pragma unroll
            for (int i = 0; i < XKEEP; ++i)
                if (tid + 256 * i < n4) quant4(tid + 256 * i, xnormw(xr[i], wr[i]));      // (n4 is a multiple of 8: whole 8-lane blocks)
        } else {
            // (rows longer than the register copy, and the plain prologue of o_proj / down_proj: four chunks' loads go out together -- one
            // L2 round trip per four chunks instead of one per chunk in front of the first dot product)
            for (int k0 = tid; k0 < n4; k0 += 1024) {
                f32x4 xb[4], wb[4], zb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k4 = k0 + 256 * i < n4 ? k0 + 256 * i : k0;
                    xb[i] = *(const f32x4*)(a.x + (k4 << 2));
                    if (PRO == PRO_RMSNORM) wb[i] = *(const f32x4*)(a.nw + (k4 << 2));
                    if (PRO == PRO_GDNNORM) { zb[i] = *(const f32x4*)(a.gdn_z + (k4 << 2)); wb[i] = *(const f32x4*)(a.gdn_w + ((k4 & 31) << 2)); }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (k0 + 256 * i >= n4) break;
                    f32x4 v = PRO == PRO_RMSNORM ? xnormw(xb[i], wb[i]) : xb[i];
                    if (PRO == PRO_GDNNORM) {
                        // gemv_bf16_kernel's PRO_GDNNORM arithmetic: a value head = the float4 of 32 consecutive lanes (K / 4 is a multiple of 32)
                        float hs = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
                        hs += __shfl_xor(hs, 1); hs += __shfl_xor(hs, 2); hs += __shfl_xor(hs, 4); hs += __shfl_xor(hs, 8); hs += __shfl_xor(hs, 16);
                        const float rms = 1.0f / sqrtf(hs / 128.0f + a.eps);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = v[e] * rms * wb[i][e] * (zb[i][e] / (1.0f + expf(-zb[i][e])));
                    }
                    quant4(k0 + 256 * i, v);
                }
            }
        }
    } else {
        // quantize_row_q8_K: one wave per 256-element block
        const int nblk = K >> 8;
        for (int blk0 = wave; blk0 < nblk; blk0 += 16) {
          // (four blocks' loads of this wave together: every block is a chain of wave reductions behind its load)
          f32x4 xb4[4], wb4[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
              const int kk = (blk0 + 4 * i < nblk ? blk0 + 4 * i : blk0) * 64 + lane;
              xb4[i] = *(const f32x4*)(a.x + (kk << 2));
              if (PRO == PRO_RMSNORM) wb4[i] = *(const f32x4*)(a.nw + (kk << 2));
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int blk = blk0 + 4 * i;
            if (blk >= nblk) break;
            const int k4 = blk * 64 + lane;
            const f32x4 v = PRO == PRO_RMSNORM ? xnormw(xb4[i], wb4[i]) : xb4[i];
            // signed value of the FIRST element with the largest |x|
            unsigned long long key = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned long long ke = ((unsigned long long)__float_as_uint(fabsf(v[e])) << 32) | (unsigned)(255 - (lane * 4 + e));
                key = ke > key ? ke : key;
            }
            unsigned long long best = key;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)best, o), hi = (unsigned)__shfl_xor((int)(unsigned)(best >> 32), o);
                const unsigned long long ot = ((unsigned long long)hi << 32) | lo;
                best = ot > best ? ot : best;
            }
            const int widx = 255 - (int)(unsigned)(best & 0xFFFFFFFFull);
            float cand = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) if (lane * 4 + e == widx) cand = v[e];
            const float mx = wave_sum(cand);
            const float iscale = mx != 0.f ? -128.0f / mx : 0.f;
            int q[4]; int s4 = 0; uint32_t pk = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                q[e] = mx != 0.f ? min(127, __float2int_rn(v[e] * iscale)) : 0;
                s4 += q[e];
                pk |= ((uint32_t)q[e] & 0xFFu) << (8 * e);
            }
            ((uint32_t*)xq)[k4] = pk;
            const int s8 = s4 + __shfl_xor(s4, 1);
            if (!(lane & 1)) xs8[k4 >> 1] = s8;
            if (lane == 0) xd[blk] = mx != 0.f ? 1.0f / iscale : 0.f;
          }
        }
    }
    __syncthreads();

    float best = -INFINITY; int besti = 0x7FFFFFFF;
    float acc[R];
#pragma unroll
    for (int i = 0; i < R; ++i) acc[i] = 0.f;
    auto dot = [&](const QRow (&q)[R], int c) {
        if constexpr (FMT == QFMT_Q8_0) {
            const int e0 = c * 1024 + lane * 16;
            const u32x4 xv = *(const u32x4*)(xq + e0);
            const float dx = xd[e0 >> 5];
#pragma unroll
            for (int i = 0; i < R; ++i) {
                int isum = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) isum = __builtin_amdgcn_sdot4((int)q[i].a[j], (int)xv[j], isum, false);
                acc[i] = fmaf(q_d(q[i]) * dx, (float)isum, acc[i]);
            }
        } else if constexpr (FMT == QFMT_Q4_K) {
            const int kb = c * 8 + (lane >> 3), h = lane & 7, p = h >> 1;
            const int e0 = kb * 256 + 64 * p + 16 * (h & 1);
            const u32x4 x0 = *(const u32x4*)(xq + e0), x1 = *(const u32x4*)(xq + e0 + 32);
            const int bs0 = xs8[e0 >> 3] + xs8[(e0 >> 3) + 1], bs1 = xs8[(e0 + 32) >> 3] + xs8[((e0 + 32) >> 3) + 1];
            const float dx = xd[kb];
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const u32x4 hb = q[i].b;
                const float d = f16_bits_to_f32(hb[0] & 0xFFFFu), dmin = f16_bits_to_f32(hb[0] >> 16);
                auto sbyte = [&](int ix) -> int { const int bi = 4 + ix; return (int)((hb[bi >> 2] >> (8 * (bi & 3))) & 0xFFu); };
                auto scale_min = [&](int j, int& sc, int& mn) {
                    if (j < 4) { sc = sbyte(j) & 63; mn = sbyte(j + 4) & 63; }
                    else { sc = (sbyte(j + 4) & 0xF) | ((sbyte(j - 4) >> 6) << 4); mn = (sbyte(j + 4) >> 4) | ((sbyte(j) >> 6) << 4); }
                };
                int sc0, m0, sc1, m1;
                scale_min(2 * p, sc0, m0);
                scale_min(2 * p + 1, sc1, m1);
                int il = 0, ih = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    il = __builtin_amdgcn_sdot4((int)(q[i].a[j] & 0x0F0F0F0Fu), (int)x0[j], il, false);
                    ih = __builtin_amdgcn_sdot4((int)((q[i].a[j] >> 4) & 0x0F0F0F0Fu), (int)x1[j], ih, false);
                }
                acc[i] = fmaf(-(dx * dmin), (float)(m0 * bs0 + m1 * bs1), fmaf(dx * d, (float)(sc0 * il + sc1 * ih), acc[i]));
            }
        } else {
            const int kb = c * 8 + (lane >> 3), n = (lane >> 2) & 1, j = lane & 3;
            const int e0 = kb * 256 + 128 * n + 8 * j;
            u32x2 xr[4]; int bs[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) { xr[t] = *(const u32x2*)(xq + e0 + 32 * t); bs[t] = xs8[(e0 + 32 * t) >> 3]; }
            const float dx = xd[kb];
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const float d = q_d(q[i]);
                int sumi = 0;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int qsel = (t & 1) * 2, hshift = 2 * t;
                    int is = 0;
#pragma unroll
                    for (int wi = 0; wi < 2; ++wi) {
                        const uint32_t qlw = q[i].a[qsel + wi], qhw = q[i].b[wi];
                        const uint32_t code = ((t < 2 ? qlw : (qlw >> 4)) & 0x0F0F0F0Fu) | (((qhw >> hshift) & 0x03030303u) << 4);
                        is = __builtin_amdgcn_sdot4((int)code, (int)xr[t][wi], is, false);
                    }
                    const int sc = q6_sc(q[i], t, lane);
                    sumi += sc * (is - 32 * bs[t]);
                }
                acc[i] = fmaf(dx * d, (float)sumi, acc[i]);
            }
        }
    };
    auto finish = [&](int g) {
        const int r0 = g * R;
        float mine = 0.f, mine_up = 0.f;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            acc[i] = wave_sum(acc[i]);
            if (EPI != EPI_SILUMUL && lane == i) mine = acc[i];
            if (EPI == EPI_SILUMUL && (i & 1) && lane == (i >> 1)) { mine = acc[i - 1]; mine_up = acc[i]; }
        }
        if (EPI == EPI_STORE) {
            if (lane < R && r0 + lane < N) a.y[r0 + lane] = mine;
        } else if (EPI == EPI_RESADD) {
            if (lane < R && r0 + lane < N) {
                if (a.res == a.y) atomicAdd(&a.y[r0 + lane], mine);          // in-place residual: no load to wait for
                else a.y[r0 + lane] = a.res[r0 + lane] + mine;
            }
        } else if (EPI == EPI_SILUMUL) {
            if (lane < R / 2 && r0 + 2 * lane + 1 < N) a.y[(r0 >> 1) + lane] = (mine / (1.0f + expf(-mine))) * mine_up;
        } else if (EPI == EPI_ARGMAX) {
            if (lane < R && r0 + lane < N) a.y[r0 + lane] = mine;
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const int ix = r0 + i + a.idx_base;
                if (r0 + i < N && (acc[i] > best || (acc[i] == best && ix < besti))) { best = acc[i]; besti = ix; }
            }
        }
#pragma unroll
        for (int i = 0; i < R; ++i) acc[i] = 0.f;
    };
    // one continuous pipeline over (row group, chunk): the next one is requested before the current one is dotted
    int g = gfirst, c = 0;
    while (g < G) {
        int ng = g, nc = c + 1;
        if (nc >= nch) { ng = g + gstride; nc = 0; }
        load_rows(qb, ng, nc);
        dot(qa, c);
        if (nc == 0) finish(g);
        g = ng; c = nc;
        if (g >= G) break;
        ng = g; nc = c + 1;
        if (nc >= nch) { ng = g + gstride; nc = 0; }
        load_rows(qa, ng, nc);
        dot(qb, c);
        if (nc == 0) finish(g);
        g = ng; c = nc;
    }
    if (EPI == EPI_ARGMAX) {
        __syncthreads();
        int* redi = (int*)(red + 4);
        if (lane == 0) { red[wave] = best; redi[wave] = besti; }
        __syncthreads();
        if (tid == 0) {
            float bb = red[0]; int bbi = redi[0];
            for (int w = 1; w < 4; ++w)
                if (red[w] > bb || (red[w] == bb && redi[w] < bbi)) { bb = red[w]; bbi = redi[w]; }
            a.pmax[blockIdx.x] = bb; a.pidx[blockIdx.x] = bbi;
        }
    }
}

template <int FMT>
static void launch_gemvq_i8_f(int pro, int epi, const GemvQArgs& a, int grid, hipStream_t s) {
    constexpr int CK = QF<FMT>::CK;
    const size_t kpad = (size_t)((a.w.K + CK - 1) / CK) * CK;
    const size_t lds = kpad + (FMT == QFMT_Q8_0 ? kpad / 32 * 4 : kpad / 256 * 4 + kpad / 8 * 4) + 64;
#define CM_QI(P, E) { hipLaunchKernelGGL((gemvq_i8_kernel<FMT, P, E>), dim3(grid), dim3(256), lds, s, a); return; }
    if (pro == PRO_RMSNORM) {
        if (epi == EPI_STORE) CM_QI(PRO_RMSNORM, EPI_STORE)
        if (epi == EPI_SILUMUL) CM_QI(PRO_RMSNORM, EPI_SILUMUL)
        if (epi == EPI_ARGMAX) CM_QI(PRO_RMSNORM, EPI_ARGMAX)
        CM_QI(PRO_RMSNORM, EPI_RESADD)
    } else if (pro == PRO_GDNNORM && FMT == QFMT_Q8_0 && (epi == EPI_RESADD || epi == EPI_STORE)) {
        if (epi == EPI_STORE) CM_QI(PRO_GDNNORM, EPI_STORE)
        CM_QI(PRO_GDNNORM, EPI_RESADD)
    } else {
        if (epi == EPI_STORE) CM_QI(PRO_PLAIN, EPI_STORE)
        if (epi == EPI_SILUMUL) CM_QI(PRO_PLAIN, EPI_SILUMUL)
        if (epi == EPI_ARGMAX) CM_QI(PRO_PLAIN, EPI_ARGMAX)
        CM_QI(PRO_PLAIN, EPI_RESADD)
    }
#undef CM_QI
}

int gemvq_grid(int N, int num_cu, int fmt) {
    // measured on Qwen3-8B Q8_0 (tok/s): 2 rows per wave 379 (4 blocks per CU) / 378 (8); 4 rows per wave 363 / 362
    const int rows = fmt == QFMT_Q8_0 ? CM_Q8_R : 2;
    const int groups = (N + rows - 1) / rows;
    return std::max(1, std::min((groups + 3) / 4, num_cu * 4));
}

template <int FMT>
static void launch_gemvq_f(int pro, int epi, const GemvQArgs& a, int grid, hipStream_t s) {
    constexpr int CK = QF<FMT>::CK;
    const size_t lds = (size_t)((a.w.K + CK - 1) / CK) * CK * 4 + 64;
#define CM_Q(P, E) { static DevOnce attr; \
        attr.run([] { (void)hipFuncSetAttribute((const void*)gemvq_kernel<FMT, P, E>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }); \
        hipLaunchKernelGGL((gemvq_kernel<FMT, P, E>), dim3(grid), dim3(256), lds, s, a); return; }
    if (pro == PRO_RMSNORM) {
        if (epi == EPI_STORE) CM_Q(PRO_RMSNORM, EPI_STORE)
        if (epi == EPI_SILUMUL) CM_Q(PRO_RMSNORM, EPI_SILUMUL)
        if (epi == EPI_ARGMAX) CM_Q(PRO_RMSNORM, EPI_ARGMAX)
        CM_Q(PRO_RMSNORM, EPI_RESADD)
    } else {
        if (epi == EPI_STORE) CM_Q(PRO_PLAIN, EPI_STORE)
        if (epi == EPI_SILUMUL) CM_Q(PRO_PLAIN, EPI_SILUMUL)
        if (epi == EPI_ARGMAX) CM_Q(PRO_PLAIN, EPI_ARGMAX)
        CM_Q(PRO_PLAIN, EPI_RESADD)
    }
#undef CM_Q
}

bool launch_gemvq(int pro, int epi, const GemvQArgs& a, int grid, hipStream_t s) {
    if (a.act_int) {
        switch (a.w.fmt) {
            case QFMT_Q8_0: launch_gemvq_i8_f<QFMT_Q8_0>(pro, epi, a, grid, s); return true;
            case QFMT_Q4_K: launch_gemvq_i8_f<QFMT_Q4_K>(pro, epi, a, grid, s); return true;
            case QFMT_Q6_K: launch_gemvq_i8_f<QFMT_Q6_K>(pro, epi, a, grid, s); return true;
            default: return false;
        }
    }
    switch (a.w.fmt) {
        case QFMT_Q8_0: launch_gemvq_f<QFMT_Q8_0>(pro, epi, a, grid, s); return true;
        case QFMT_Q4_K: launch_gemvq_f<QFMT_Q4_K>(pro, epi, a, grid, s); return true;
        case QFMT_Q6_K: launch_gemvq_f<QFMT_Q6_K>(pro, epi, a, grid, s); return true;
        default: return false;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Batched integer-dot GEMV (continuous batching over quantised weights, qwen3/modeling.rs:1202-1234 with
// LinearLayer::Quantized): up to MB sequences share ONE pass over the weight stream.  The weight bytes of a
// (row, chunk) are decoded once and dotted with every sequence's activation codes, so the per-weight decode
// cost that bounds gemvq_i8_kernel is amortised over the batch.  Each sequence's row is quantised exactly as
// gemvq_i8_kernel does it (two 256-thread halves of the block take alternate sequences and walk K with the same
// thread -> element mapping), and a wave sums a row in the same order, so y[m] is bit-identical to the
// single-sequence kernel on x[m].
//   block = 512 threads (8 waves); LDS per sequence = Kpad codes + block scales (+ Kpad/8 code sums for K-quants)
// ---------------------------------------------------------------------------------------------------------
template <int FMT, int PRO, int EPI, int MB>
__global__ __launch_bounds__(512, 2) void gemvqb_i8_kernel(GemvQBArgs a0) {
    using F = QF<FMT>;
    // 9..64 sequences: 2 / 4 / 8 GROUPS of <= 8 share the stream of codes through the L2, exactly as the bf16 matrix-core
    // GEMV does (kernels_decode_mfma.hip): block ids (8 ngrp) k + 8 g + j = logical block 8 k + j of group g, same XCD
    GemvQBArgs a = a0;
    const int lgq = a0.n_seq > 32 ? 3 : (a0.n_seq > 16 ? 2 : (a0.n_seq > 8 ? 1 : 0)), ngq = 1 << lgq;
    const int gq = ((int)blockIdx.x >> 3) & (ngq - 1);
    const int lblk = ngq > 1 ? ((int)blockIdx.x >> (3 + lgq)) * 8 + ((int)blockIdx.x & 7) : (int)blockIdx.x;
    const int nlb = (int)gridDim.x >> lgq;
    if (ngq > 1) {
        a.x = a0.x + (size_t)gq * 8 * a0.ldx;
        a.y = a0.y + (size_t)gq * 8 * a0.ldy;
        a.res = a0.res == a0.y ? a.y : (a0.res != nullptr ? a0.res + (size_t)gq * 8 * a0.ldy : nullptr);
        a.n_seq = max(0, min(8, a0.n_seq - gq * 8));
        if (a.n_seq == 0) {                                   // padding group
            if (EPI == EPI_ARGMAX && (int)threadIdx.x < a0.n_seq) {
                a0.pmax[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = -INFINITY;
                a0.pidx[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = 0x7FFFFFFF;
            }
            return;
        }
    }
    constexpr bool KQ = FMT != QFMT_Q8_0;
    // rows per wave: 4 for Q8_0 up to 4 sequences (halves the LDS reads per weight), 2 otherwise (measured: Q8_0 at 8
    // sequences 5.95 ms/step with 2 rows vs 6.67 with 4; the K-quants sit at the 256-VGPR budget with 2).  U chunks per
    // load batch stays 1: two chunks in flight spill (R = 4) or gain nothing (the kernel is issue-bound, not
    // bytes-in-flight-bound).
    constexpr int R = (!KQ && MB <= 4) ? 4 : 2, U = 1, CK = F::CK, NW = 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = a.w.K, N = a.w.N, ns = a.n_seq;
    const int nch = (K + CK - 1) / CK;
    const int Kpad = nch * CK;
    const int nscale = KQ ? Kpad / 256 : Kpad / 32;
    const int seq_bytes = Kpad + nscale * 4 + (KQ ? Kpad / 8 * 4 : 0);
    float* red = (float*)(lds_raw + (size_t)MB * seq_bytes);       // [MB][4] sums of squares, later [NW][MB] arg-max pairs

    const int lane_k = (FMT == QFMT_Q8_0) ? lane * 16 : (lane >> 3) * 256;
    const int G = (N + R - 1) / R;
    const int gstride = nlb * NW;
    const int gfirst = lblk * NW + wave;
    QRow q[R][U], qn[R][U];
    // unconditional, clamped loads (see gemvq_i8_kernel): lanes / steps past K or N re-read in-bounds bytes whose
    // product is zero (codes, scales and code sums are zero past K) or that are never consumed
    auto load_rows = [&](QRow (&dst)[R][U], int g, int c0) {
        const int r0 = min(g, G - 1) * R;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool inr = c0 + u < nch && (c0 + u) * CK + lane_k < K;
            const int ce = inr ? c0 + u : 0, le = inr ? lane : (KQ ? (lane & 7) : 0);
#pragma unroll
            for (int i = 0; i < R; ++i) dst[i][u] = q_load<FMT>(a.w, min(r0 + i, N - 1), ce, le);
        }
    };
    load_rows(q, gfirst, 0);                       // weight bytes are requested before the activation rows are quantised

    // ---- quantise the activation rows: half `grp` of the block takes sequences grp, grp + 2, ... ----
    // Loads are issued SG sequences x 4 strides at a time: one L2 round trip per 1024 elements of K for all of a
    // half-block's sequences (a load-per-iteration loop costs a round trip each: ~16 us per launch at 8 x 4096).
    constexpr int SG = MB / 2;
    const int grp = tid >> 8, t2 = tid & 255, w2 = wave & 3;
    const int n4 = K >> 2;
    auto xrow = [&](int mi) { return a.x + (size_t)min(grp + 2 * mi, ns - 1) * a.ldx; };   // clamped: loads stay in-bounds
    if (PRO == PRO_RMSNORM) {
        float ss[SG];
#pragma unroll
        for (int mi = 0; mi < SG; ++mi) ss[mi] = 0.f;
        for (int kb = t2; kb < n4; kb += 1024) {
            f32x4 v[SG][4];
#pragma unroll
            for (int mi = 0; mi < SG; ++mi)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    v[mi][j] = (kb + 256 * j < n4) ? *(const f32x4*)(xrow(mi) + ((kb + 256 * j) << 2)) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int mi = 0; mi < SG; ++mi)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (kb + 256 * j < n4)
                        ss[mi] = fmaf(v[mi][j][3], v[mi][j][3], fmaf(v[mi][j][2], v[mi][j][2], fmaf(v[mi][j][1], v[mi][j][1], fmaf(v[mi][j][0], v[mi][j][0], ss[mi]))));
        }
#pragma unroll
        for (int mi = 0; mi < SG; ++mi) {
            const float t = wave_sum(ss[mi]);
            if (lane == 0 && grp + 2 * mi < ns) red[(grp + 2 * mi) * 4 + w2] = t;
        }
        __syncthreads();
    }
    float rr[SG];
#pragma unroll
    for (int mi = 0; mi < SG; ++mi) {
        rr[mi] = 1.f;
        const int m = min(grp + 2 * mi, ns - 1);
        if (PRO == PRO_RMSNORM)
            rr[mi] = 1.0f / sqrtf(((red[m * 4] + red[m * 4 + 1]) + (red[m * 4 + 2] + red[m * 4 + 3])) / (float)K + a.eps);
    }
    auto normed = [&](f32x4 v, const f32x4& w, float r) -> f32x4 {
        if (PRO == PRO_RMSNORM) {
            v[0] = __fmul_rn(__fmul_rn(v[0], r), w[0]); v[1] = __fmul_rn(__fmul_rn(v[1], r), w[1]);
            v[2] = __fmul_rn(__fmul_rn(v[2], r), w[2]); v[3] = __fmul_rn(__fmul_rn(v[3], r), w[3]);
        }
        return v;
    };
    if (K < Kpad) {                                                // lanes past K multiply zeros (see load_rows)
        for (int m = 0; m < ns; ++m) {
            unsigned char* sp = lds_raw + (size_t)m * seq_bytes;
            float* zd = (float*)(sp + Kpad);
            int* zs = (int*)(zd + nscale);
            for (int e = (K >> 2) + tid; e < (Kpad >> 2); e += 512) ((uint32_t*)sp)[e] = 0;
            for (int e = (KQ ? K >> 8 : K >> 5) + tid; e < nscale; e += 512) zd[e] = 0.f;
            if (KQ) for (int e = (K >> 3) + tid; e < (Kpad >> 3); e += 512) zs[e] = 0;
        }
    }
    if (!KQ) {
        for (int kb = t2; kb < n4; kb += 1024) {                   // quantize_row_q8_0: 32 elements = 8 consecutive lanes
            f32x4 v[SG][4], nwv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool live = kb + 256 * j < n4;
                nwv[j] = (PRO == PRO_RMSNORM && live) ? *(const f32x4*)(a.nw + ((kb + 256 * j) << 2)) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int mi = 0; mi < SG; ++mi)
                    v[mi][j] = live ? *(const f32x4*)(xrow(mi) + ((kb + 256 * j) << 2)) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int mi = 0; mi < SG; ++mi) {
                const int m = grp + 2 * mi;
                if (m >= ns) break;
                signed char* xq = (signed char*)(lds_raw + (size_t)m * seq_bytes);
                float* xd = (float*)(xq + Kpad);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k4 = kb + 256 * j;
                    if (k4 >= n4) break;
                    const f32x4 x = normed(v[mi][j], nwv[j], rr[mi]);
                    float am = fmaxf(fmaxf(fabsf(x[0]), fabsf(x[1])), fmaxf(fabsf(x[2]), fabsf(x[3])));
                    am = fmaxf(am, __shfl_xor(am, 1)); am = fmaxf(am, __shfl_xor(am, 2)); am = fmaxf(am, __shfl_xor(am, 4));
                    const float d = am / 127.0f;
                    const float id = d != 0.f ? 1.0f / d : 0.f;
                    uint32_t pk = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) pk |= ((uint32_t)(int)roundf(x[e] * id) & 0xFFu) << (8 * e);
                    ((uint32_t*)xq)[k4] = pk;
                    if ((t2 & 7) == 0) xd[k4 >> 3] = f16_round(d);
                }
            }
        }
    } else {
        const int nblk = K >> 8;                                   // quantize_row_q8_K: one wave per 256-element block
        for (int b0 = w2; b0 < nblk; b0 += 16) {
            f32x4 v[SG][4], nwv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool live = b0 + 4 * j < nblk;
                const int k4 = (b0 + 4 * j) * 64 + lane;
                nwv[j] = (PRO == PRO_RMSNORM && live) ? *(const f32x4*)(a.nw + (k4 << 2)) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int mi = 0; mi < SG; ++mi)
                    v[mi][j] = live ? *(const f32x4*)(xrow(mi) + (k4 << 2)) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int mi = 0; mi < SG; ++mi) {
                const int m = grp + 2 * mi;
                if (m >= ns) break;
                signed char* xq = (signed char*)(lds_raw + (size_t)m * seq_bytes);
                float* xd = (float*)(xq + Kpad);
                int* xs8 = (int*)(xd + nscale);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int blk = b0 + 4 * j;
                    if (blk >= nblk) break;
                    const int k4 = blk * 64 + lane;
                    const f32x4 x = normed(v[mi][j], nwv[j], rr[mi]);
                    unsigned long long key = 0;                    // signed value of the FIRST element with the largest |x|
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned long long ke = ((unsigned long long)__float_as_uint(fabsf(x[e])) << 32) | (unsigned)(255 - (lane * 4 + e));
                        key = ke > key ? ke : key;
                    }
                    unsigned long long bestk = key;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) {
                        const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)bestk, o), hi = (unsigned)__shfl_xor((int)(unsigned)(bestk >> 32), o);
                        const unsigned long long ot = ((unsigned long long)hi << 32) | lo;
                        bestk = ot > bestk ? ot : bestk;
                    }
                    const int widx = 255 - (int)(unsigned)(bestk & 0xFFFFFFFFull);
                    float cand = 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (lane * 4 + e == widx) cand = x[e];
                    const float mx = wave_sum(cand);
                    const float iscale = mx != 0.f ? -128.0f / mx : 0.f;
                    int qq[4]; int s4 = 0; uint32_t pk = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        qq[e] = mx != 0.f ? min(127, __float2int_rn(x[e] * iscale)) : 0;
                        s4 += qq[e];
                        pk |= ((uint32_t)qq[e] & 0xFFu) << (8 * e);
                    }
                    ((uint32_t*)xq)[k4] = pk;
                    const int s8 = s4 + __shfl_xor(s4, 1);
                    if (!(lane & 1)) xs8[k4 >> 1] = s8;
                    if (lane == 0) xd[blk] = mx != 0.f ? 1.0f / iscale : 0.f;
                }
            }
        }
    }
    __syncthreads();

    float best[MB]; int besti[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) { best[m] = -INFINITY; besti[m] = 0x7FFFFFFF; }
    for (int g = gfirst; g < G; g += gstride) {
        const int r0 = g * R;
        float acc[R][MB];
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
            for (int m = 0; m < MB; ++m) acc[i][m] = 0.f;
        for (int c0 = 0; c0 < nch; c0 += U) {
            // the next batch of (row group, chunks) is in flight while this one is dotted with every sequence
            if (c0 + U < nch) load_rows(qn, g, c0 + U);
            else load_rows(qn, g + gstride, 0);                    // past the last row group: a clamped, unused re-read
#pragma unroll
            for (int u = 0; u < U; ++u) {
            const int c = c0 + u;
            {
                if constexpr (FMT == QFMT_Q8_0) {
                    const int e0 = c * 1024 + lane * 16;
                    float dw[R];
#pragma unroll
                    for (int i = 0; i < R; ++i) dw[i] = q_d(q[i][u]);
#pragma unroll
                    for (int m = 0; m < MB; ++m) {
                        if (m >= ns) break;
                        const unsigned char* sp = lds_raw + (size_t)m * seq_bytes;
                        const u32x4 xv = *(const u32x4*)(sp + e0);
                        const float dx = ((const float*)(sp + Kpad))[e0 >> 5];
#pragma unroll
                        for (int i = 0; i < R; ++i) {
                            int isum = 0;
#pragma unroll
                            for (int j = 0; j < 4; ++j) isum = __builtin_amdgcn_sdot4((int)q[i][u].a[j], (int)xv[j], isum, false);
                            acc[i][m] = fmaf(dw[i] * dx, (float)isum, acc[i][m]);
                        }
                    }
                } else if constexpr (FMT == QFMT_Q4_K) {
                    const int kb = c * 8 + (lane >> 3), h = lane & 7, p = h >> 1;
                    const int e0 = kb * 256 + 64 * p + 16 * (h & 1);
                    float d[R], dmin[R]; int sc0[R], m0[R], sc1[R], m1[R]; u32x4 wl[R], wh[R];
#pragma unroll
                    for (int i = 0; i < R; ++i) {
                        const u32x4 hb = q[i][u].b;
                        d[i] = f16_bits_to_f32(hb[0] & 0xFFFFu); dmin[i] = f16_bits_to_f32(hb[0] >> 16);
                        auto sbyte = [&](int ix) -> int { const int bi = 4 + ix; return (int)((hb[bi >> 2] >> (8 * (bi & 3))) & 0xFFu); };
                        auto scale_min = [&](int j, int& sc, int& mn) {
                            if (j < 4) { sc = sbyte(j) & 63; mn = sbyte(j + 4) & 63; }
                            else { sc = (sbyte(j + 4) & 0xF) | ((sbyte(j - 4) >> 6) << 4); mn = (sbyte(j + 4) >> 4) | ((sbyte(j) >> 6) << 4); }
                        };
                        scale_min(2 * p, sc0[i], m0[i]);
                        scale_min(2 * p + 1, sc1[i], m1[i]);
#pragma unroll
                        for (int j = 0; j < 4; ++j) { wl[i][j] = q[i][u].a[j] & 0x0F0F0F0Fu; wh[i][j] = (q[i][u].a[j] >> 4) & 0x0F0F0F0Fu; }
                    }
#pragma unroll
                    for (int m = 0; m < MB; ++m) {
                        if (m >= ns) break;
                        const unsigned char* sp = lds_raw + (size_t)m * seq_bytes;
                        const float* xd = (const float*)(sp + Kpad);
                        const int* xs8 = (const int*)(xd + nscale);
                        const u32x4 x0 = *(const u32x4*)(sp + e0), x1 = *(const u32x4*)(sp + e0 + 32);
                        const int bs0 = xs8[e0 >> 3] + xs8[(e0 >> 3) + 1], bs1 = xs8[(e0 + 32) >> 3] + xs8[((e0 + 32) >> 3) + 1];
                        const float dx = xd[kb];
#pragma unroll
                        for (int i = 0; i < R; ++i) {
                            int il = 0, ih = 0;
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                il = __builtin_amdgcn_sdot4((int)wl[i][j], (int)x0[j], il, false);
                                ih = __builtin_amdgcn_sdot4((int)wh[i][j], (int)x1[j], ih, false);
                            }
                            acc[i][m] = fmaf(-(dx * dmin[i]), (float)(m0[i] * bs0 + m1[i] * bs1), fmaf(dx * d[i], (float)(sc0[i] * il + sc1[i] * ih), acc[i][m]));
                        }
                    }
                } else {
                    const int kb = c * 8 + (lane >> 3), n = (lane >> 2) & 1, j = lane & 3;
                    const int e0 = kb * 256 + 128 * n + 8 * j;
                    float d[R]; uint32_t code[R][4][2]; int sc[R][4];
#pragma unroll
                    for (int i = 0; i < R; ++i) {
                        d[i] = q_d(q[i][u]);
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const int qsel = (t & 1) * 2, hshift = 2 * t;
#pragma unroll
                            for (int wi = 0; wi < 2; ++wi) {
                                const uint32_t qlw = q[i][u].a[qsel + wi], qhw = q[i][u].b[wi];
                                code[i][t][wi] = ((t < 2 ? qlw : (qlw >> 4)) & 0x0F0F0F0Fu) | (((qhw >> hshift) & 0x03030303u) << 4);
                            }
                            sc[i][t] = q6_sc(q[i][u], t, lane);
                        }
                    }
#pragma unroll
                    for (int m = 0; m < MB; ++m) {
                        if (m >= ns) break;
                        const unsigned char* sp = lds_raw + (size_t)m * seq_bytes;
                        const float* xd = (const float*)(sp + Kpad);
                        const int* xs8 = (const int*)(xd + nscale);
                        u32x2 xr[4]; int bs[4];
#pragma unroll
                        for (int t = 0; t < 4; ++t) { xr[t] = *(const u32x2*)(sp + e0 + 32 * t); bs[t] = xs8[(e0 + 32 * t) >> 3]; }
                        const float dx = xd[kb];
#pragma unroll
                        for (int i = 0; i < R; ++i) {
                            int sumi = 0;
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                int is = 0;
#pragma unroll
                                for (int wi = 0; wi < 2; ++wi) is = __builtin_amdgcn_sdot4((int)code[i][t][wi], (int)xr[t][wi], is, false);
                                sumi += sc[i][t] * (is - 32 * bs[t]);
                            }
                            acc[i][m] = fmaf(dx * d[i], (float)sumi, acc[i][m]);
                        }
                    }
                }
            }
            if (U > 1) __builtin_amdgcn_sched_barrier(0);          // keep one chunk's activation codes live at a time (VGPR budget)
            }
#pragma unroll
            for (int i = 0; i < R; ++i)
#pragma unroll
                for (int u = 0; u < U; ++u) q[i][u] = qn[i][u];
        }
        // lane m*R + i ends up holding (row r0 + i, sequence m); SiLU*mul: lane m*(R/2) + p holds the (gate, up) pair p
        float mine = 0.f, gate_v = 0.f, up_v = 0.f;
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            if (m >= ns) break;
#pragma unroll
            for (int i = 0; i < R; ++i) {
                acc[i][m] = wave_sum(acc[i][m]);
                if (lane == m * R + i) mine = acc[i][m];
            }
            if (EPI == EPI_SILUMUL) {
#pragma unroll
                for (int pp = 0; pp < R / 2; ++pp)
                    if (lane == m * (R / 2) + pp) { gate_v = acc[2 * pp][m]; up_v = acc[2 * pp + 1][m]; }
            }
            if (EPI == EPI_ARGMAX) {
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    const int ix = r0 + i + a.idx_base;
                    if (r0 + i < N && (acc[i][m] > best[m] || (acc[i][m] == best[m] && ix < besti[m]))) { best[m] = acc[i][m]; besti[m] = ix; }
                }
            }
        }
        const int mi = lane / R, ri = lane % R;
        if (EPI == EPI_STORE || EPI == EPI_ARGMAX) {
            if (mi < ns && r0 + ri < N) a.y[(size_t)mi * a.ldy + r0 + ri] = mine;
        } else if (EPI == EPI_RESADD) {
            if (mi < ns && r0 + ri < N) {
                if (a.res == a.y) atomicAdd(&a.y[(size_t)mi * a.ldy + r0 + ri], mine);      // in-place residual: no load to wait for
                else a.y[(size_t)mi * a.ldy + r0 + ri] = a.res[(size_t)mi * a.ldy + r0 + ri] + mine;
            }
        } else if (EPI == EPI_SILUMUL) {
            const int ms = lane / (R / 2), ps = lane % (R / 2);
            if (ms < ns && r0 + 2 * ps + 1 < N) a.y[(size_t)ms * a.ldy + (r0 >> 1) + ps] = (gate_v / (1.0f + expf(-gate_v))) * up_v;
        }
    }
    if (EPI == EPI_ARGMAX) {
        __syncthreads();
        int* redi = (int*)(red + NW * MB);
        if (lane == 0) {
#pragma unroll
            for (int m = 0; m < MB; ++m) { red[wave * MB + m] = best[m]; redi[wave * MB + m] = besti[m]; }
        }
        __syncthreads();
        if (tid < ns) {
            float bb = red[tid]; int bbi = redi[tid];
            for (int w = 1; w < NW; ++w)
                if (red[w * MB + tid] > bb || (red[w * MB + tid] == bb && redi[w * MB + tid] < bbi)) { bb = red[w * MB + tid]; bbi = redi[w * MB + tid]; }
            a.pmax[(size_t)(gq * 8 + tid) * gridDim.x + blockIdx.x] = bb; a.pidx[(size_t)(gq * 8 + tid) * gridDim.x + blockIdx.x] = bbi;
        }
        // several groups: this block's column of the OTHER groups' rows must not win (argmax_final scans every column)
        if (ngq > 1 && tid < ngq * 8 && (tid >> 3) != gq && tid < a0.n_seq) {
            a.pmax[(size_t)tid * gridDim.x + blockIdx.x] = -INFINITY;
            a.pidx[(size_t)tid * gridDim.x + blockIdx.x] = 0x7FFFFFFF;
        }
    }
}

static size_t gemvqb_seq_bytes(int fmt, int K) {
    const int CK = fmt == QFMT_Q8_0 ? 1024 : 2048;
    const size_t kpad = (size_t)((K + CK - 1) / CK) * CK;
    return kpad + (fmt == QFMT_Q8_0 ? kpad / 32 * 4 : kpad / 256 * 4 + kpad / 8 * 4);
}
constexpr size_t QB_LDS_MAX = 160 * 1024 - 1024;

// sequences one launch can take (0: the activation codes of even one sequence pair do not fit in LDS)
int gemvqb_max_seqs(int fmt, int K) {
    const size_t fit = (QB_LDS_MAX - 8 * 8 * 8) / gemvqb_seq_bytes(fmt, K);
    return fit >= 8 ? 8 : fit >= 4 ? 4 : 0;
}
int gemvqb_grid(int fmt, int N, int K, int n_seq, int num_cu) {
    (void)K;
    // every block re-quantises the activation rows and the K-quant kernels hold > 128 VGPRs: one fat block per CU
    const int rows = (fmt == QFMT_Q8_0 && n_seq <= 4) ? 4 : 2;
    const int groups = (N + rows - 1) / rows;
    if (n_seq <= 8) return std::max(1, std::min((groups + 7) / 8, num_cu));
    // 2 / 4 / 8 sequence groups: 1 / ngrp of the CUs each, logical blocks per group a multiple of 8 (the id mapping)
    const int ngrp = n_seq > 32 ? 8 : (n_seq > 16 ? 4 : 2);
    int per = std::max(1, std::min((groups + 7) / 8, std::max(8, num_cu / ngrp)));
    per = (per + 7) / 8 * 8;
    return ngrp * per;
}

template <int FMT, int MB>
static void launch_gemvqb_t(int pro, int epi, const GemvQBArgs& a, int grid, hipStream_t s) {
    const size_t lds = (size_t)MB * gemvqb_seq_bytes(FMT, a.w.K) + 8 * 8 * 8;
#define CM_QB(P, E) { static DevOnce attr; \
        attr.run([] { (void)hipFuncSetAttribute((const void*)gemvqb_i8_kernel<FMT, P, E, MB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }); \
        hipLaunchKernelGGL((gemvqb_i8_kernel<FMT, P, E, MB>), dim3(grid), dim3(512), lds, s, a); return; }
    if (pro == PRO_RMSNORM && epi == EPI_STORE) CM_QB(PRO_RMSNORM, EPI_STORE)
    if (pro == PRO_RMSNORM && epi == EPI_SILUMUL) CM_QB(PRO_RMSNORM, EPI_SILUMUL)
    if (pro == PRO_RMSNORM && epi == EPI_ARGMAX) CM_QB(PRO_RMSNORM, EPI_ARGMAX)
    if (pro == PRO_PLAIN && epi == EPI_RESADD) CM_QB(PRO_PLAIN, EPI_RESADD)
#undef CM_QB
    throw std::runtime_error("gemvqb: prologue/epilogue combination not built");
}

// n_seq <= gemvqb_max_seqs(fmt, K); the (prologue, epilogue) pairs the decoder uses
bool launch_gemvqb(int pro, int epi, const GemvQBArgs& a, int grid, hipStream_t s) {
    if (a.n_seq < 1 || a.n_seq > 64 || (a.n_seq > 8 && gemvqb_max_seqs(a.w.fmt, a.w.K) < 8)) return false;
#define CM_QBF(F) { if (a.n_seq <= 4) launch_gemvqb_t<F, 4>(pro, epi, a, grid, s); else launch_gemvqb_t<F, 8>(pro, epi, a, grid, s); return true; }
    switch (a.w.fmt) {
        case QFMT_Q8_0: CM_QBF(QFMT_Q8_0)
        case QFMT_Q4_K: CM_QBF(QFMT_Q4_K)
        case QFMT_Q6_K: CM_QBF(QFMT_Q6_K)
        default: return false;
    }
#undef CM_QBF
}

// ---------------------------------------------------------------------------------------------------------
// element-wise dequantisation (embedding-row gather; also the reference for the GEMV in tests)
// ---------------------------------------------------------------------------------------------------------
__device__ float q_elem(const QWeight& w, size_t row, int k) {
    const size_t K = (size_t)w.K;
    if (w.fmt == QFMT_Q8_0) {
        const float d = f16_bits_to_f32(*(const uint16_t*)(w.p1 + (row * (K >> 5) + (size_t)(k >> 5)) * 2));
        return d * (float)(int)((const signed char*)w.p0)[row * K + (size_t)k];
    }
    const size_t blk = row * (K >> 8) + (size_t)(k >> 8);
    const int e = k & 255;
    if (w.fmt == QFMT_Q4_K) {
        const uint8_t* h = w.p1 + blk * 16;
        const float d = f16_bits_to_f32((uint32_t)h[0] | ((uint32_t)h[1] << 8)), dmin = f16_bits_to_f32((uint32_t)h[2] | ((uint32_t)h[3] << 8));
        const uint8_t* sc = h + 4;
        const int j = e >> 5;
        uint32_t s, m;
        if (j < 4) { s = sc[j] & 63u; m = sc[j + 4] & 63u; }
        else { s = (sc[j + 4] & 0xFu) | ((uint32_t)(sc[j - 4] >> 6) << 4); m = (uint32_t)(sc[j + 4] >> 4) | ((uint32_t)(sc[j] >> 6) << 4); }
        const uint8_t byte = w.p0[blk * 128 + (size_t)(e >> 6) * 32 + (e & 31)];
        const uint32_t q = ((e >> 5) & 1) ? (byte >> 4) : (byte & 0xFu);
        return (d * (float)s) * (float)q - (dmin * (float)m);
    }
    // Q6_K
    const int n = e >> 7, f = e & 127, t = f >> 5, l = f & 31;
    const uint8_t qlb = w.p0[blk * 128 + (size_t)n * 64 + (t & 1) * 32 + l];
    const uint8_t qhb = w.p1[blk * 64 + (size_t)n * 32 + l];
    const uint32_t q = ((t < 2) ? (qlb & 0xFu) : (uint32_t)(qlb >> 4)) | (((uint32_t)(qhb >> (2 * t)) & 3u) << 4);
    const float sc = (float)(int)((const signed char*)w.p2)[blk * 16 + (size_t)n * 8 + (l >> 4) + 2 * t];
    const float d = f16_bits_to_f32(*(const uint16_t*)(w.p3 + blk * 2));
    return (d * sc) * (float)((int)q - 32);
}

__global__ void embed_row_q_kernel(QWeight w, const StepState* __restrict__ st, float* __restrict__ x, int H, int V) {
    st += blockIdx.y;                              // batched step: one state / output row per sequence
    x += (size_t)blockIdx.y * H;
    uint32_t tok = st->token;
    if (tok >= (uint32_t)V) tok = 0;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < H) x[i] = q_elem(w, (size_t)tok, i);
}

__global__ void dequant_rows_kernel(QWeight w, float* __restrict__ out, int row0, int nrows) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)nrows * w.K) return;
    const size_t r = i / (size_t)w.K;
    out[i] = q_elem(w, (size_t)row0 + r, (int)(i % (size_t)w.K));
}

// prefill over quantised weights: rows of W are dequantised to a bf16 scratch matrix for the MFMA GEMM (bf16 rounding
// of the dequantised value, 2^-9, is below every format's own quantisation step); dst row = row * row_mul + row_off
// so gate / up matrices of different ggml types can still be interleaved
// out_lo (parity mode): the second bf16 term of the dequantised value, bf16(v - bf16(v)) -- an 8-bit code times an f16 scale (K-quants:
// times a 6-bit sub-scale, minus a min) has up to ~25 significant bits; hi + lo carries 16 of them (2^-17), hi alone 8 (2^-9)
__global__ void dequant_bf16_kernel(QWeight w, uint16_t* __restrict__ out, uint16_t* __restrict__ out_lo, int row_mul, int row_off) {
    const size_t i8 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i8 >= (size_t)w.N * w.K) return;
    const size_t r = i8 / (size_t)w.K;
    const int k = (int)(i8 % (size_t)w.K);
    uint32_t o[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float v0 = q_elem(w, r, k + 2 * e), v1 = q_elem(w, r, k + 2 * e + 1);
        const uint16_t h0 = f32_to_bf16(v0), h1 = f32_to_bf16(v1);
        o[e] = (uint32_t)h0 | ((uint32_t)h1 << 16);
        l[e] = (uint32_t)f32_to_bf16(v0 - bf16_to_f32(h0)) | ((uint32_t)f32_to_bf16(v1 - bf16_to_f32(h1)) << 16);
    }
    const size_t at = (r * row_mul + row_off) * (size_t)w.K + k;
    *(u32x4*)(out + at) = (u32x4){o[0], o[1], o[2], o[3]};
    if (out_lo != nullptr) *(u32x4*)(out_lo + at) = (u32x4){l[0], l[1], l[2], l[3]};
}
void launch_dequant_bf16(const QWeight& w, uint16_t* out, int row_mul, int row_off, hipStream_t s, uint16_t* out_lo) {
    const size_t n8 = (size_t)w.N * w.K / 8;
    hipLaunchKernelGGL(dequant_bf16_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, w, out, out_lo, row_mul, row_off);
}

__global__ void embed_rows_q_kernel(QWeight w, const uint32_t* __restrict__ ids, float* __restrict__ x, int H, int V) {
    uint32_t tok = ids[blockIdx.x];
    if (tok >= (uint32_t)V) tok = 0;
    for (int i = threadIdx.x; i < H; i += blockDim.x) x[(size_t)blockIdx.x * H + i] = q_elem(w, (size_t)tok, i);
}
void launch_embed_rows_q(const QWeight& w, const uint32_t* ids, float* x, int S, int H, int V, hipStream_t s) {
    hipLaunchKernelGGL(embed_rows_q_kernel, dim3(S), dim3(256), 0, s, w, ids, x, H, V);
}

void launch_embed_row_q(const QWeight& w, const StepState* st, float* x, int H, int V, hipStream_t s, int n_seq) {
    hipLaunchKernelGGL(embed_row_q_kernel, dim3((H + 255) / 256, n_seq), dim3(256), 0, s, w, st, x, H, V);
}
void launch_dequant_rows(const QWeight& w, float* out, int row0, int nrows, hipStream_t s) {
    const size_t n = (size_t)nrows * w.K;
    hipLaunchKernelGGL(dequant_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, out, row0, nrows);
}

// ---------------------------------------------------------------------------------------------------------
// in-situ quantisation bf16 -> Q8_0 (ops/linear.rs:83-100 quantize_linear; ggml quantize_row_q8_0_ref:
// d = amax / 127, id = d ? 1/d : 0, q = roundf(x * id), d stored as f16)
// ---------------------------------------------------------------------------------------------------------
// MODE 8: Q8_0.  MODE 4 / 5: ggml quantize_row_q4_0_ref / _q5_0_ref (max = the signed value of the first largest |x|, d = max / -8 | -16,
// q = min(15 | 31, (int8)(x * id + 8.5 | 16.5)), d stored as f16); the code q - 8 | 16 is an int8 under the block's own scale, i.e. the
// block is stored in the Q8_0 stream layout (loader_gguf.cpp pack_rows does the same to Q4_0 / Q5_0 files)
template <int MODE>
__global__ void isq_kernel(const uint16_t* __restrict__ src, size_t src_stride, int N, int K, signed char* __restrict__ codes,
                           uint16_t* __restrict__ dd) {
    // the reference computes x * id, THEN adds 8.5 / 16.5 and truncates: a fused multiply-add lands on the other side of an integer
    // for weights that sit exactly on a code boundary (x / d = -4.5: seen on hardware, one weight in ~10^4).  hipcc contracts by
    // default and __fmul_rn / __fadd_rn are plain operators in this ROCm (clang/22/include/__clang_hip_math.h:271), so say it here
#pragma clang fp contract(off)
    const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // one 32-weight block per thread
    const size_t nb_row = (size_t)K >> 5;
    if (b >= (size_t)N * nb_row) return;
    const size_t row = b / nb_row, kb = b % nb_row;
    const uint16_t* p = src + row * src_stride + kb * 32;
    float v[32];
    float amax = 0.f, mx = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        v[i] = bf16_to_f32(p[i]);
        if (amax < fabsf(v[i])) { amax = fabsf(v[i]); mx = v[i]; }
    }
    const float d = MODE == 8 ? amax / 127.0f : MODE == 4 ? mx / -8.0f : mx / -16.0f;
    const float id = d != 0.f ? 1.0f / d : 0.f;
    signed char* q = codes + row * (size_t)K + kb * 32;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        if (MODE == 8) q[i] = (signed char)(int)roundf(v[i] * id);
        else if (MODE == 4) { const float t = v[i] * id; q[i] = (signed char)(min(15, (int)(signed char)(t + 8.5f)) - 8); }
        else { const float t = v[i] * id; q[i] = (signed char)(min(31, (int)(signed char)(t + 16.5f)) - 16); }
    }
    const _Float16 h = (_Float16)d;
    uint16_t hb;
    __builtin_memcpy(&hb, &h, 2);
    dd[b] = hb;
}

void launch_isq_q8_0(const uint16_t* src, size_t src_stride, int N, int K, void* codes, void* d, hipStream_t s, int mode) {
    const size_t nb = (size_t)N * (K >> 5);
    const dim3 g((unsigned)((nb + 127) / 128)), b(128);
    if (mode == 4) hipLaunchKernelGGL(isq_kernel<4>, g, b, 0, s, src, src_stride, N, K, (signed char*)codes, (uint16_t*)d);
    else if (mode == 5) hipLaunchKernelGGL(isq_kernel<5>, g, b, 0, s, src, src_stride, N, K, (signed char*)codes, (uint16_t*)d);
    else hipLaunchKernelGGL(isq_kernel<8>, g, b, 0, s, src, src_stride, N, K, (signed char*)codes, (uint16_t*)d);
}

}  // namespace cm
