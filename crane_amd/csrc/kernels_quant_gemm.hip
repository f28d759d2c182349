// Decode groups over quantised weights on the int8 matrix cores (round 4; SURVEY 8f rank 3, the review's "int8-MFMA GEMM for
// quantised decode groups").  The batched integer-dot GEMV (gemvqb_i8_kernel, kernels_quant.hip) decodes a weight row once per
// <= 8 sequences on the VALU (v_dot4): a 64-sequence round of Qwen3-8B in Q8_0 ran 2.6 K tok/s against 7.4 K on bf16 weights.
// Here a group of up to 128 sequences is ONE pass over the codes:
//
//   quant_rows_q8_kernel : every activation row -> Q8_0 blocks (32 codes + an f16-rounded scale), the arithmetic of the GEMV
//                          kernels' prologue bit for bit (RMSNorm sums in the same order, d = amax / 127, roundf(x / d)) --
//                          once per projection input, not once per workgroup;
//   gemm_q8_i8_kernel    : C[m][n] = sum over 32-element blocks b of (dw[n][b] * dx[m][b]) * (int32 dot of the block's codes) --
//                          ggml_vec_dot_q8_0_q8_0 per (m, n): the integer dots on v_mfma_i32_32x32x16_i8 (two per block: a lane's
//                          16 bytes of a row are k 16 h .. 16 h + 15 of the block, low half -> first MFMA, high half -> second;
//                          weights and activations use the same map), the block scales applied on the VALU to the 16 int32 a lane
//                          gets per (m-tile, block).  A wave owns 32 weight rows and streams them straight into registers (a
//                          weight byte is read by exactly one wave); the activation codes (M x K bytes, L2-resident) come the
//                          same way, their block scales ride along, transposed ([block][row]).
//                          Split K over workgroups to fill the chip; f32 partials [ks][M][N];
//   q8_splitk_epilogue   : adds the slices in order and stores / adds the residual / SiLU(gate) * up -- f32 rows, the input of the
//                          next projection's quantiser.
//
// Same products as the integer-dot GEMVs (same codes, same scales, exact int32 dots); the f32 sums are taken per 32-block in k
// order instead of per lane chunk, so rows agree with the single-sequence step to rounding of the f32 sums (and to the code
// flips that rounding can cause one projection later), not bit for bit -- which is why small groups stay on the GEMV
// (cm_debug_set("q_gemm_min")).  Weight layout: QFMT_Q8_0 (GGUF Q8_0 / Q4_0 / Q5_0 tensors and the ISQ modes, all widened to
// it at load); K-quants stay on the GEMV path.
#include <algorithm>
#include <cstdlib>

#include "dev_common.h"
#include "kernels.h"

namespace cm {

namespace {

typedef int i32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float f16r(float v) { const _Float16 h = (_Float16)v; return (float)h; }
__device__ __forceinline__ float f16bits(uint32_t b) { _Float16 v; const uint16_t u = (uint16_t)b; __builtin_memcpy(&v, &u, 2); return (float)v; }

}  // namespace

// one 256-thread block per row: thread t walks k4 = t + 256 j + 1024 i exactly like a 256-thread half of gemvqb_i8_kernel
template <bool NORM>
__global__ __launch_bounds__(256) void quant_rows_q8_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ nw, float eps,
                                                            signed char* __restrict__ xq, float* __restrict__ xd, int K) {
    __shared__ float red[4];
    const int m = blockIdx.x, t2 = threadIdx.x, lane = t2 & 63, w2 = t2 >> 6;
    const float* xr = x + (size_t)m * ldx;
    const int n4 = K >> 2;
    float r = 1.f;
    if (NORM) {
        float ss = 0.f;
        for (int kb = t2; kb < n4; kb += 1024) {
            f32x4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (kb + 256 * j < n4) ? *(const f32x4*)(xr + ((kb + 256 * j) << 2)) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (kb + 256 * j < n4) ss = fmaf(v[j][3], v[j][3], fmaf(v[j][2], v[j][2], fmaf(v[j][1], v[j][1], fmaf(v[j][0], v[j][0], ss))));
        }
        const float t = wave_sum(ss);
        if (lane == 0) red[w2] = t;
        __syncthreads();
        r = 1.0f / sqrtf(((red[0] + red[1]) + (red[2] + red[3])) / (float)K + eps);
    }
    uint32_t* xqr = (uint32_t*)(xq + (size_t)m * K);
    for (int kb = t2; kb < n4; kb += 1024) {
        f32x4 v[4], nwv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool live = kb + 256 * j < n4;
            nwv[j] = (NORM && live) ? *(const f32x4*)(nw + ((kb + 256 * j) << 2)) : (f32x4){0.f, 0.f, 0.f, 0.f};
            v[j] = live ? *(const f32x4*)(xr + ((kb + 256 * j) << 2)) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k4 = kb + 256 * j;
            if (k4 >= n4) break;
            f32x4 xv = v[j];
            if (NORM) {
                xv[0] = __fmul_rn(__fmul_rn(xv[0], r), nwv[j][0]); xv[1] = __fmul_rn(__fmul_rn(xv[1], r), nwv[j][1]);
                xv[2] = __fmul_rn(__fmul_rn(xv[2], r), nwv[j][2]); xv[3] = __fmul_rn(__fmul_rn(xv[3], r), nwv[j][3]);
            }
            float am = fmaxf(fmaxf(fabsf(xv[0]), fabsf(xv[1])), fmaxf(fabsf(xv[2]), fabsf(xv[3])));
            am = fmaxf(am, __shfl_xor(am, 1)); am = fmaxf(am, __shfl_xor(am, 2)); am = fmaxf(am, __shfl_xor(am, 4));
            const float d = am / 127.0f;
            const float id = d != 0.f ? 1.0f / d : 0.f;
            uint32_t pk = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk |= ((uint32_t)(int)roundf(xv[e] * id) & 0xFFu) << (8 * e);
            xqr[k4] = pk;
            if ((t2 & 7) == 0) xd[(size_t)(k4 >> 3) * QGEMM_MAXM + m] = f16r(d);
        }
    }
}

void launch_quant_rows_q8(const float* x, int ldx, const float* nw, float eps, signed char* xq, float* xd, int M, int K, hipStream_t s) {
    if (nw) hipLaunchKernelGGL(quant_rows_q8_kernel<true>, dim3(M), dim3(256), 0, s, x, ldx, nw, eps, xq, xd, K);
    else hipLaunchKernelGGL(quant_rows_q8_kernel<false>, dim3(M), dim3(256), 0, s, x, ldx, nw, eps, xq, xd, K);
}

// MT m-tiles of 32 rows (M <= 32 MT); grid = (N / 128) * ksplit workgroups of 4 waves, wave w owns weight rows [128 tn + 32 w, + 32).
// K is walked in GROUPS of QG = 8 blocks (256 codes per row):
//   * weights: a lane's 8 x 16 bytes of the NEXT group and its 8 block scales (one 16-byte load) are requested while the current
//     group is multiplied -- 8 KB per wave one group (~2 us of work) ahead; a first version with one block in flight ran 0.6 TB/s;
//   * activation codes: the group's panel (M rows x 256 bytes) is fetched ONCE per workgroup (16 bytes x 2 MT per thread, into
//     registers one group ahead, then into one of two LDS panels, the group's block scales next to it), the four waves read their
//     fragments from there -- loaded by every wave straight from the L2 they were 4 x the weight bytes.  78 KB of LDS at 128 rows:
//     two workgroups per CU, one's MFMA latency under the other's scaling (the head at 128 rows: 605 -> 412 us);
//     (measured, wrong results by construction: the same bytes as 1 KB-contiguous wave loads instead of 32 rows x 32 bytes are
//     only 25 % faster -- the kernel is bound by the VALU work of the scaling, not by its load pattern);
//   * scaling: 16 int32 per lane and (m-tile, block) -> f32, times dw[n] * dx[m], as packed f32 math (v_pk_mul / v_pk_fma).
constexpr int QG = 8;
constexpr int QROWB = QG * 32 + 16;                 // bytes of a panel row in LDS (padded: rows 16 bytes apart in the banks)

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MT>
__global__ __launch_bounds__(256, 2) void gemm_q8_i8_kernel(QGemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char qlds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int K = a.w.K, N = a.w.N, nkb_all = K >> 5, nkb = nkb_all / a.ksplit, ngrp = nkb / QG;
    float* xds = (float*)qlds;                                              // [2][QG][QGEMM_MAXM] block scales of the activation rows, per group
    unsigned char* As = qlds + (size_t)2 * QG * QGEMM_MAXM * sizeof(float);  // [2][32 MT][QROWB] activation codes of a group
    constexpr int PANEL = MT * 32 * QROWB;
    const int tiles = N / 128;
    const int ks = (int)blockIdx.x / tiles, tn = (int)blockIdx.x % tiles;
    const int kb0 = ks * nkb;
    const int n = tn * 128 + wave * 32 + r;                                 // this lane's weight row = its output column
    const uint8_t* wp = a.w.p0 + (size_t)n * K + (size_t)kb0 * 32 + 16 * h;
    const uint16_t* dp = (const uint16_t*)a.w.p1 + (size_t)n * nkb_all + kb0;
    // activation panel: chunk c = tid + 256 i (i < 2 MT): row c / 16, 16-byte chunk c % 16 of the group's 256 bytes
    const signed char* xsrc[2 * MT];
    int xdst[2 * MT];
#pragma unroll
    for (int i = 0; i < 2 * MT; ++i) {
        const int c = tid + 256 * i, row = c >> 4, q = c & 15;
        xsrc[i] = a.xq + (size_t)min(row, a.M - 1) * K + (size_t)kb0 * 32 + 16 * q;       // (rows past M: clamped, dropped at the store)
        xdst[i] = row * QROWB + 16 * q;
    }
    u32x4 wv[QG], wn[QG], sc, scn, areg[2 * MT];
#pragma unroll
    for (int j = 0; j < QG; ++j) wv[j] = ld_nt16(wp + j * 32);
    sc = *(const u32x4*)dp;
#pragma unroll
    for (int i = 0; i < 2 * MT; ++i) areg[i] = *(const u32x4*)xsrc[i];
    // (a group's scales: QG x 128 floats = one 16-byte load per thread)
    const f32x4* xdsrc = (const f32x4*)(a.xd + (size_t)kb0 * QGEMM_MAXM) + tid;
    f32x4 xreg = *xdsrc;
    ((f32x4*)xds)[tid] = xreg;
#pragma unroll
    for (int i = 0; i < 2 * MT; ++i) *(u32x4*)(As + xdst[i]) = areg[i];
    f32x2 acc[MT][8];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[mt][i] = (f32x2){0.f, 0.f};
    __syncthreads();
    for (int g = 0; g < ngrp; ++g) {
        const bool more = g + 1 < ngrp;
        const int gn = more ? g + 1 : g;                                    // (the last group re-reads itself: unused)
#pragma unroll
        for (int j = 0; j < QG; ++j) wn[j] = ld_nt16(wp + (size_t)(gn * QG + j) * 32);
        scn = *(const u32x4*)(dp + gn * QG);
#pragma unroll
        for (int i = 0; i < 2 * MT; ++i) areg[i] = *(const u32x4*)(xsrc[i] + (size_t)gn * QG * 32);
        xreg = xdsrc[(size_t)gn * (QG * QGEMM_MAXM / 4)];
        const unsigned char* Ap = As + (g & 1) * PANEL;
        const float* xg = xds + (g & 1) * (QG * QGEMM_MAXM);
#pragma unroll
        for (int j = 0; j < QG; ++j) {
            const float dw = f16bits((sc[j >> 1] >> (16 * (j & 1))) & 0xFFFFu);
            const long wlo = (long)(((unsigned long)wv[j][1] << 32) | wv[j][0]), whi = (long)(((unsigned long)wv[j][3] << 32) | wv[j][2]);
            const float* xr = xg + j * QGEMM_MAXM + 4 * h;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const u32x4 av = *(const u32x4*)(Ap + (mt * 32 + r) * QROWB + j * 32 + 16 * h);
                const long alo = (long)(((unsigned long)av[1] << 32) | av[0]), ahi = (long)(((unsigned long)av[3] << 32) | av[2]);
                i32x16 c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                c = __builtin_amdgcn_mfma_i32_32x32x16_i8(alo, wlo, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_i32_32x32x16_i8(ahi, whi, c, 0, 0, 0);
                // D[i] of lane (r, h): row 8 (i / 4) + 4 h + i % 4 of the m-tile, column r
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 dx = *(const f32x4*)(xr + mt * 32 + 8 * q);
                    const f32x2 s0 = (f32x2){dw, dw} * (f32x2){dx[0], dx[1]}, s1 = (f32x2){dw, dw} * (f32x2){dx[2], dx[3]};
                    const f32x2 c0 = (f32x2){(float)c[4 * q], (float)c[4 * q + 1]}, c1 = (f32x2){(float)c[4 * q + 2], (float)c[4 * q + 3]};
                    acc[mt][2 * q] = __builtin_elementwise_fma(s0, c0, acc[mt][2 * q]);
                    acc[mt][2 * q + 1] = __builtin_elementwise_fma(s1, c1, acc[mt][2 * q + 1]);
                }
                // one m-tile's int32 results and scales live at a time: left alone, the MFMAs and LDS reads of a whole group are issued
                // first and their conversions sink below all of them (> 512 registers).  The empty asm is ordered like any volatile
                // statement and needs the sums, so the scaling of this m-tile stays in front of the next one's MFMAs.
#pragma unroll
                for (int q = 0; q < 8; ++q) asm volatile("" : "+v"(acc[mt][q]));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (more) {
            unsigned char* An = As + ((g + 1) & 1) * PANEL;
#pragma unroll
            for (int i = 0; i < 2 * MT; ++i) *(u32x4*)(An + xdst[i]) = areg[i];
            ((f32x4*)(xds + ((g + 1) & 1) * (QG * QGEMM_MAXM)))[tid] = xreg;
        }
        __syncthreads();
        sc = scn;
#pragma unroll
        for (int j = 0; j < QG; ++j) wv[j] = wn[j];
    }
    float* P = a.ws + (size_t)ks * a.slice;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = mt * 32 + 8 * (i >> 2) + 4 * h + (i & 3);
            if (m < a.M) P[(size_t)m * a.ldp + n] = acc[mt][i >> 1][i & 1];
        }
}

// slices added in the order 0, 1, ...; 4 consecutive columns of a row per thread.  EPI_STORE / EPI_RESADD: y[m][n] (+)= v;
// EPI_SILUMUL: columns (gate_j, up_j) interleaved -> y[m][n / 2] = silu(gate) * up (the GEMV epilogue's expression)
template <int EPI, int KS>
__global__ __launch_bounds__(256) void q8_splitk_epilogue_kernel(const float* __restrict__ ws, int M, int N, int ksplit, float* __restrict__ y, int ldy) {
    const size_t n4 = (size_t)N / 4, total = (size_t)M * n4, slice = (size_t)M * N;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
        const int m = (int)(t / n4), n = (int)(t % n4) * 4;
        const float* p0 = ws + (size_t)m * N + n;
        f32x4 v;
        if (KS > 0) {
            f32x4 p[KS > 0 ? KS : 1];
#pragma unroll
            for (int k = 0; k < KS; ++k) p[k] = *(const f32x4*)(p0 + (size_t)k * slice);
            v = p[0];
#pragma unroll
            for (int k = 1; k < KS; ++k) { v[0] += p[k][0]; v[1] += p[k][1]; v[2] += p[k][2]; v[3] += p[k][3]; }
        } else {
            v = *(const f32x4*)p0;
            for (int k = 1; k < ksplit; ++k) {
                const f32x4 p = *(const f32x4*)(p0 + (size_t)k * slice);
                v[0] += p[0]; v[1] += p[1]; v[2] += p[2]; v[3] += p[3];
            }
        }
        if (EPI == EPI_SILUMUL) {
            float* o = y + (size_t)m * ldy + (n >> 1);
            o[0] = (v[0] / (1.0f + expf(-v[0]))) * v[1];
            o[1] = (v[2] / (1.0f + expf(-v[2]))) * v[3];
        } else {
            float* o = y + (size_t)m * ldy + n;
            if (EPI == EPI_RESADD) { const f32x4 c = *(const f32x4*)o; v[0] += c[0]; v[1] += c[1]; v[2] += c[2]; v[3] += c[3]; }
            *(f32x4*)o = v;
        }
    }
}

// The reduction + the NEXT projection's quantiser in one launch: one 256-thread block per row, thread t owns the float4 chunks
// k4 = t + 256 j + 1024 i of the OUTPUT row like quant_rows_q8_kernel, so codes and scales are bit-equal to the two launches.
//   EPI_RESADD : y[m][n] += sum of the slices (N columns), then RMSNorm (NORM) + Q8_0 blocks of the row
//   EPI_SILUMUL: y[m][k] = silu(gate_k) * up_k from the interleaved columns (2 k, 2 k + 1) (N / 2 outputs), then Q8_0 blocks
template <int EPI, int KS, bool NORM>
__global__ __launch_bounds__(256) void q8_splitk_quant_kernel(const float* __restrict__ ws, int M, int N, int ksplit, float* __restrict__ y, int ldy,
                                                              const float* __restrict__ nw, float eps, signed char* __restrict__ xq, float* __restrict__ xd) {
    __shared__ float red[4];
    const int m = blockIdx.x, t2 = threadIdx.x, lane = t2 & 63, w2 = t2 >> 6;
    const int Kout = EPI == EPI_SILUMUL ? N / 2 : N, n4 = Kout >> 2;
    const size_t slice = (size_t)M * N;
    float* yr = y + (size_t)m * ldy;
    auto reduce4 = [&](int col) -> f32x4 {                        // 4 consecutive columns of the row, slices in order
        const float* p0 = ws + (size_t)m * N + col;
        f32x4 v;
        if (KS > 0) {
            f32x4 p[KS > 0 ? KS : 1];
#pragma unroll
            for (int k = 0; k < KS; ++k) p[k] = *(const f32x4*)(p0 + (size_t)k * slice);
            v = p[0];
#pragma unroll
            for (int k = 1; k < KS; ++k) { v[0] += p[k][0]; v[1] += p[k][1]; v[2] += p[k][2]; v[3] += p[k][3]; }
        } else {
            v = *(const f32x4*)p0;
            for (int k = 1; k < ksplit; ++k) {
                const f32x4 p = *(const f32x4*)(p0 + (size_t)k * slice);
                v[0] += p[0]; v[1] += p[1]; v[2] += p[2]; v[3] += p[3];
            }
        }
        return v;
    };
    // pass 1: the row of y (what q8_splitk_epilogue_kernel writes)
    for (int k4 = t2; k4 < n4; k4 += 256) {
        f32x4 o;
        if (EPI == EPI_SILUMUL) {
            const f32x4 a0 = reduce4(8 * k4), a1 = reduce4(8 * k4 + 4);
            o[0] = (a0[0] / (1.0f + expf(-a0[0]))) * a0[1]; o[1] = (a0[2] / (1.0f + expf(-a0[2]))) * a0[3];
            o[2] = (a1[0] / (1.0f + expf(-a1[0]))) * a1[1]; o[3] = (a1[2] / (1.0f + expf(-a1[2]))) * a1[3];
        } else {
            const f32x4 v = reduce4(4 * k4), c = *(const f32x4*)(yr + 4 * k4);
            o = (f32x4){v[0] + c[0], v[1] + c[1], v[2] + c[2], v[3] + c[3]};
        }
        *(f32x4*)(yr + 4 * k4) = o;
    }
    // pass 2 + 3: quant_rows_q8_kernel over this thread's own stores
    float r = 1.f;
    if (NORM) {
        float ss = 0.f;
        for (int kb = t2; kb < n4; kb += 1024) {
            f32x4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (kb + 256 * j < n4) ? *(const f32x4*)(yr + ((kb + 256 * j) << 2)) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (kb + 256 * j < n4) ss = fmaf(v[j][3], v[j][3], fmaf(v[j][2], v[j][2], fmaf(v[j][1], v[j][1], fmaf(v[j][0], v[j][0], ss))));
        }
        const float t = wave_sum(ss);
        if (lane == 0) red[w2] = t;
        __syncthreads();
        r = 1.0f / sqrtf(((red[0] + red[1]) + (red[2] + red[3])) / (float)Kout + eps);
    }
    uint32_t* xqr = (uint32_t*)(xq + (size_t)m * Kout);
    for (int kb = t2; kb < n4; kb += 1024) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k4 = kb + 256 * j;
            if (k4 >= n4) break;
            f32x4 xv = *(const f32x4*)(yr + (k4 << 2));
            if (NORM) {
                const f32x4 w = *(const f32x4*)(nw + (k4 << 2));
                xv[0] = __fmul_rn(__fmul_rn(xv[0], r), w[0]); xv[1] = __fmul_rn(__fmul_rn(xv[1], r), w[1]);
                xv[2] = __fmul_rn(__fmul_rn(xv[2], r), w[2]); xv[3] = __fmul_rn(__fmul_rn(xv[3], r), w[3]);
            }
            float am = fmaxf(fmaxf(fabsf(xv[0]), fabsf(xv[1])), fmaxf(fabsf(xv[2]), fabsf(xv[3])));
            am = fmaxf(am, __shfl_xor(am, 1)); am = fmaxf(am, __shfl_xor(am, 2)); am = fmaxf(am, __shfl_xor(am, 4));
            const float d = am / 127.0f;
            const float id = d != 0.f ? 1.0f / d : 0.f;
            uint32_t pk = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk |= ((uint32_t)(int)roundf(xv[e] * id) & 0xFFu) << (8 * e);
            xqr[k4] = pk;
            if ((t2 & 7) == 0) xd[(size_t)(k4 >> 3) * QGEMM_MAXM + m] = f16r(d);
        }
    }
}

template <int EPI>
static void launch_q8_epilogue_quant(const float* ws, int M, int N, int ks, float* y, int ldy, const QNext& nx, hipStream_t s) {
#define CM_QQ(KS_) do { if (nx.nw) hipLaunchKernelGGL((q8_splitk_quant_kernel<EPI, KS_, true>), dim3(M), dim3(256), 0, s, ws, M, N, ks, y, ldy, nx.nw, nx.eps, nx.xq, nx.xd); \
                        else hipLaunchKernelGGL((q8_splitk_quant_kernel<EPI, KS_, false>), dim3(M), dim3(256), 0, s, ws, M, N, ks, y, ldy, nx.nw, nx.eps, nx.xq, nx.xd); } while (0)
    if (ks == 1) CM_QQ(1); else if (ks == 2) CM_QQ(2); else if (ks == 4) CM_QQ(4); else if (ks == 8) CM_QQ(8); else if (ks == 16) CM_QQ(16); else CM_QQ(0);
#undef CM_QQ
}

template <int EPI>
static void launch_q8_epilogue(const float* ws, int M, int N, int ks, float* y, int ldy, hipStream_t s) {
    const int eb = (int)std::min<size_t>(((size_t)M * (N / 4) + 255) / 256, 2048);
    if (ks == 1) hipLaunchKernelGGL((q8_splitk_epilogue_kernel<EPI, 1>), dim3(eb), dim3(256), 0, s, ws, M, N, ks, y, ldy);
    else if (ks == 2) hipLaunchKernelGGL((q8_splitk_epilogue_kernel<EPI, 2>), dim3(eb), dim3(256), 0, s, ws, M, N, ks, y, ldy);
    else if (ks == 4) hipLaunchKernelGGL((q8_splitk_epilogue_kernel<EPI, 4>), dim3(eb), dim3(256), 0, s, ws, M, N, ks, y, ldy);
    else if (ks == 8) hipLaunchKernelGGL((q8_splitk_epilogue_kernel<EPI, 8>), dim3(eb), dim3(256), 0, s, ws, M, N, ks, y, ldy);
    else if (ks == 16) hipLaunchKernelGGL((q8_splitk_epilogue_kernel<EPI, 16>), dim3(eb), dim3(256), 0, s, ws, M, N, ks, y, ldy);
    else hipLaunchKernelGGL((q8_splitk_epilogue_kernel<EPI, 0>), dim3(eb), dim3(256), 0, s, ws, M, N, ks, y, ldy);
}

bool gemm_q8_ok(const QWeight& w, int M) {
    return w.fmt == QFMT_Q8_0 && M >= 1 && M <= QGEMM_MAXM && w.N % 128 == 0 && w.K % (32 * QG) == 0;
}

// y (+)= dequant(W) . dequant(xq)^T over the group's rows; epi = EPI_STORE | EPI_RESADD | EPI_SILUMUL (GEMV epilogue codes).
// EPI_STORE with a row stride the workspace cannot hold (the vocabulary head) is written in place by an unsplit launch.
bool launch_gemm_q8(const QGemmArgs& a0, int epi, float* y, int ldy, float* ws, size_t ws_floats, int num_cu, hipStream_t s,
                    const QNext* next, bool* fused) {
    QGemmArgs a = a0;
    if (fused) *fused = false;
    if (!gemm_q8_ok(a.w, a.M) || (epi != EPI_STORE && epi != EPI_RESADD && epi != EPI_SILUMUL)) return false;
    const int N = a.w.N, nkb_all = a.w.K >> 5, tiles = N / 128;
    // split K until the chip is full (~2 workgroups per CU), the scales of a workgroup's k range fit 64 KB of LDS and the partials
    // fit the workspace
    int ks = 1;
    const int mt = (a.M + 31) / 32;
    // (two activation panels + two sets of block scales: independent of the split)
    const size_t lds = (size_t)2 * QG * QGEMM_MAXM * sizeof(float) + (size_t)2 * mt * 32 * QROWB, lds_max = 160 * 1024;
    if (lds > lds_max) return false;
    while (ks < 16 && nkb_all % (ks * 2 * QG) == 0 && tiles * ks < 2 * num_cu && (size_t)(ks * 2) * a.M * N <= ws_floats) ks *= 2;
    const bool direct = epi == EPI_STORE && ((size_t)a.M * N > ws_floats || ks == 1);
    // the partial slices of a residual / SiLU*mul projection always go through the workspace: refuse what it cannot hold (the
    // caller falls back to the batched GEMV) instead of writing past it
    if (!direct && (ws == nullptr || (size_t)ks * a.M * N > ws_floats)) return false;
    if (direct) { ks = 1; a.ws = y; a.ldp = ldy; a.slice = 0; }
    else { a.ws = ws; a.ldp = N; a.slice = (size_t)a.M * N; }
    static const int ks_env = getenv("CM_QGEMM_KS") ? atoi(getenv("CM_QGEMM_KS")) : 0;           // tuning: force the split
    if (ks_env > 0 && !direct && nkb_all % (ks_env * QG) == 0 && (size_t)ks_env * a.M * N <= ws_floats) ks = ks_env;
    a.ksplit = ks;
    static DevOnce attr;
    attr.run([&] {
        (void)hipFuncSetAttribute((const void*)gemm_q8_i8_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)gemm_q8_i8_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)gemm_q8_i8_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)gemm_q8_i8_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    const dim3 grid(tiles * ks), block(256);
    if (mt == 1) hipLaunchKernelGGL(gemm_q8_i8_kernel<1>, grid, block, lds, s, a);
    else if (mt == 2) hipLaunchKernelGGL(gemm_q8_i8_kernel<2>, grid, block, lds, s, a);
    else if (mt == 3) hipLaunchKernelGGL(gemm_q8_i8_kernel<3>, grid, block, lds, s, a);
    else hipLaunchKernelGGL(gemm_q8_i8_kernel<4>, grid, block, lds, s, a);
    if (direct) return true;
    // the next projection's quantiser rides on the reduction launch (CM_QGEMM_QFUSE = 0: its own launch, A/B); the quantiser's
    // lane map needs whole 32-blocks per 8 lanes: output rows of a multiple of 32 columns
    static const int qfuse_env = getenv("CM_QGEMM_QFUSE") ? atoi(getenv("CM_QGEMM_QFUSE")) : 1;
    const int kout = epi == EPI_SILUMUL ? N / 2 : N;
    if (next != nullptr && qfuse_env != 0 && (epi == EPI_RESADD || epi == EPI_SILUMUL) && kout % 32 == 0) {
        if (epi == EPI_RESADD) launch_q8_epilogue_quant<EPI_RESADD>(ws, a.M, N, ks, y, ldy, *next, s);
        else launch_q8_epilogue_quant<EPI_SILUMUL>(ws, a.M, N, ks, y, ldy, *next, s);
        if (fused) *fused = true;
        return true;
    }
    if (epi == EPI_STORE) launch_q8_epilogue<EPI_STORE>(ws, a.M, N, ks, y, ldy, s);
    else if (epi == EPI_RESADD) launch_q8_epilogue<EPI_RESADD>(ws, a.M, N, ks, y, ldy, s);
    else launch_q8_epilogue<EPI_SILUMUL>(ws, a.M, N, ks, y, ldy, s);
    return true;
}

}  // namespace cm
