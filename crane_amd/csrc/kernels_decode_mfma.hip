// Batched decode on the matrix cores: y[m, n] = epilogue( W[n, :] . prologue(x[m, :]) ) for up to 8 sequences with the
// weight stream read ONCE.  The VALU formulation (kernels_decode_batch.hip) spends 8 FMAs per weight at batch 8 and is
// VALU/LDS-bound at ~2x the single-sequence time; here the sequences are the M rows of an MFMA tile, so the arithmetic
// is free and the kernel is a pure weight streamer again:
//   16 independent 4x4x4 products per instruction (mfma_f32_4x4x4bf16_1k): lane l = 4 * b + i feeds block b with
//   4 k-values of weight row i (A) and of sequence j = l % 4 (B); block b is a DIFFERENT K chunk, so one 16-byte load
//   per lane reads 4 weight rows x 256 contiguous bytes per wave instruction (the 16x16x32 tile needs 16 rows x 64 B
//   per instruction, 8 KiB apart: measured 2.5x slower, it serialises on HBM channels) and the 16 partial sums per
//   (row, sequence) are folded with four shuffles at the end of the row group
//   A = 4 weight rows straight from HBM (non-temporal), 8 x 16 B per lane in flight, 8 waves per block
//   B = the activations as bf16 hi + lo (so x keeps ~2^-17 relative precision) staged once per block in LDS
//       ([8 sequences][K slice]; the 16 lanes of one sequence read 256 contiguous bytes: conflict-free)
// A block owns ONE 4096-wide K slice of x and grid-strides over 16-row groups; K > 4096 (down_proj) splits K across
// blocks and accumulates with f32 atomics onto the residual / a zeroed buffer (store).  Same fused RMSNorm prologue
// and store / residual / SiLU*mul / arg-max epilogues as the other decode GEMVs.
// Replaces step_batch_decode's batched matmuls (reference qwen3/modeling.rs:1202-1234) for 2..8 sequences.
#include <cstdlib>

#include "dev_common.h"
#include "kernels.h"

namespace cm {

constexpr int GM_KT = 4096;          // K slice per block (elements)
constexpr int GM_MB = 8;             // sequence rows kept in LDS
constexpr int GM_LD = GM_KT + 32;    // LDS row stride (elements): +64 B, so the 4 sequences x 4 K-chunks a quarter-wave reads hit 16 distinct 16-byte bank slots
constexpr int GM_W = 8;              // waves per block (1 block per CU; 2 x 8 weight loads in flight per lane)
constexpr int GM_U = 8;              // k-steps (16-byte weight loads) in flight per lane
#ifndef GM_NT
#define GM_NT 1      // non-temporal weight loads (plain loads: 8 sequences 1575 -> 1539 tok/s, 16 sequences 2152 -> 2184)
#endif

template <int PRO, int EPI, bool TWO>
__global__ __launch_bounds__(512, 1) void gemvm_kernel(GemvBArgs a, int nkt) {
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    uint16_t* xh = lds;                                  // [GM_MB][GM_LD] (rows >= n_seq are zero)
    uint16_t* xl = lds + GM_MB * GM_LD;
    float* fsc = (float*)(xl + GM_MB * GM_LD);     // [GM_MB] 1/rms per sequence, then arg-max scratch
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = a.K, N = a.N;
    // 9..64 sequences: 2, 4 or 8 GROUPS of <= 8 share the weight stream through the L2 -- block ids (8 ngrp) k + 8 g + j (j < 8)
    // are the logical block 8 k + j of group g: the blocks that stream the same weight rows are dispatched back to back onto
    // the SAME XCD (block b runs on XCD b % 8), so the later readers of a line find it in that XCD's L2 and HBM is read once
    // (measured through the engine, Qwen3-8B: 16 running sequences 1574 -> 2152 tok/s)
    const int lg = a.n_seq > 4 * GM_MB ? 3 : (a.n_seq > 2 * GM_MB ? 2 : (a.n_seq > GM_MB ? 1 : 0)), ngrp = 1 << lg;
    const int grp = ((int)blockIdx.x >> 3) & (ngrp - 1);
    const int lblk = ((int)blockIdx.x >> (3 + lg)) * 8 + ((int)blockIdx.x & 7);
    const int nlb = (int)gridDim.x >> lg;
    const int slice = lblk % nkt, bgrp = lblk / nkt, nbg = nlb / nkt;
    const int nseq = max(0, min(GM_MB, a.n_seq - grp * GM_MB));      // sequences of this block's group (0: a padding group)
    if (nseq == 0) {                                        // padding group (e.g. 20 sequences = 8 + 8 + 4 + 0): nothing to do
        if (EPI == EPI_ARGMAX && tid < a.n_seq) {
            a.pmax[(size_t)tid * gridDim.x + blockIdx.x] = -INFINITY;
            a.pidx[(size_t)tid * gridDim.x + blockIdx.x] = 0x7FFFFFFF;
        }
        return;
    }
    const float* xg = a.x + (size_t)grp * GM_MB * a.ldx;
    float* yg = a.y + (size_t)grp * GM_MB * a.ldy;
    const float* resg = a.res != nullptr ? a.res + (size_t)grp * GM_MB * a.ldy : nullptr;
    const int k0 = slice * GM_KT;
    const int kt = min(GM_KT, K - k0);                   // multiple of 32

    // ---- stage this block's K slice of x as bf16 hi + lo; RMSNorm statistics over the WHOLE row ----
    float ss[GM_MB];
#pragma unroll
    for (int m = 0; m < GM_MB; ++m) ss[m] = 0.f;
    // 16 independent 16-byte loads per thread (the whole [8][4096] slice) are issued before the first conversion (one L2 round trip instead
    // of one per element: the un-batched loop cost ~15 us per launch)
    constexpr int SB = 16;
    static_assert((GM_MB * (GM_KT / 4)) % (64 * GM_W * SB) == 0, "staging batches");
    for (int e0 = tid; e0 < GM_MB * (GM_KT / 4); e0 += 64 * GM_W * SB) {
        f32x4 v[SB], w[SB];
#pragma unroll
        for (int i = 0; i < SB; ++i) {
            const int e = e0 + i * 64 * GM_W;
            const int m = e / (GM_KT / 4), k = (e % (GM_KT / 4)) * 4;
            const bool live = m < nseq && k < kt;
            v[i] = live ? *(const f32x4*)(xg + (size_t)m * a.ldx + k0 + k) : (f32x4){0.f, 0.f, 0.f, 0.f};
            if (PRO == PRO_RMSNORM) w[i] = live ? *(const f32x4*)(a.nw + k0 + k) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < SB; ++i) {
            const int e = e0 + i * 64 * GM_W;
            const int m = e / (GM_KT / 4), k = (e % (GM_KT / 4)) * 4;
            f32x4 x4 = v[i];
            if (PRO == PRO_RMSNORM) {
                if (nkt == 1) {          // the slice IS the row: statistics from the same pass
                    const float s2 = x4[0] * x4[0] + x4[1] * x4[1] + x4[2] * x4[2] + x4[3] * x4[3];
#pragma unroll
                    for (int mm = 0; mm < GM_MB; ++mm) if (mm == m) ss[mm] += s2;
                }
                x4[0] *= w[i][0]; x4[1] *= w[i][1]; x4[2] *= w[i][2]; x4[3] *= w[i][3];
            }
            const uint32_t h01 = pack_bf16x2(x4[0], x4[1]), h23 = pack_bf16x2(x4[2], x4[3]);
            const uint32_t l01 = pack_bf16x2(x4[0] - bf16_lo(h01), x4[1] - bf16_hi(h01));
            const uint32_t l23 = pack_bf16x2(x4[2] - bf16_lo(h23), x4[3] - bf16_hi(h23));
            *(u32x2*)&xh[m * GM_LD + k] = (u32x2){h01, h23};
            *(u32x2*)&xl[m * GM_LD + k] = (u32x2){l01, l23};
        }
    }
    if (PRO == PRO_RMSNORM) {
        if (nkt > 1)
        for (int e = tid; e < nseq * (K / 4); e += 64 * GM_W) {
            const int m = e / (K / 4), k = (e % (K / 4)) * 4;
            const f32x4 v = *(const f32x4*)(xg + (size_t)m * a.ldx + k);
            const float s2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
#pragma unroll
            for (int mm = 0; mm < GM_MB; ++mm) if (mm == m) ss[mm] += s2;
        }
        float* red = fsc + GM_MB;                        // [GM_W][GM_MB]
#pragma unroll
        for (int m = 0; m < GM_MB; ++m) {
            const float s2 = wave_sum(ss[m]);
            if (lane == 0) red[wave * GM_MB + m] = s2;
        }
        __syncthreads();
        if (tid < GM_MB) {
            float tot = 0.f;
            for (int w = 0; w < GM_W; ++w) tot += red[w * GM_MB + tid];
            fsc[tid] = 1.0f / sqrtf(tot / (float)K + a.eps);
        }
    } else if (tid < GM_MB) {
        fsc[tid] = 1.0f;
    }
    __syncthreads();

    // lane = 4 * blk + i: weight row i of the 4-row group (A) / sequence j = i (+4 for the second half) (B); block blk
    // owns the 8 k-values [128 * ks + 8 * blk, +8) of every 128-wide k-step
    const int i4 = lane & 3, blk = lane >> 2;
    const uint16_t* xh0 = xh + i4 * GM_LD + blk * 8;         // sequences 0..3
    const uint16_t* xh1 = xh + (4 + i4) * GM_LD + blk * 8;   // sequences 4..7
    const uint16_t* xl0 = xl + i4 * GM_LD + blk * 8;
    const uint16_t* xl1 = xl + (4 + i4) * GM_LD + blk * 8;
    const float sc0 = fsc[i4], sc1 = fsc[4 + i4];
    float best0 = -INFINITY, best1 = -INFINITY; int besti0 = 0x7FFFFFFF, besti1 = 0x7FFFFFFF;
    constexpr bool two = TWO;                                // sequences 4..7 present (compile-time: no branch splits the MFMA stream)

    const int G = (N + 3) / 4;
    const int nks = (kt + 127) / 128;                        // 128-wide k-steps in this slice
    const int rg_stride = nbg * GM_W;
    // 4 independent accumulators per sequence half (k-half x hi/lo): no MFMA waits on the previous one
    f32x4 c0[4], c1[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { c0[q] = (f32x4){0.f, 0.f, 0.f, 0.f}; c1[q] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    u32x4 wa[GM_U], wb[GM_U];
    // Loads are UNCONDITIONAL: a k-step past the slice re-reads the row's first bytes (x is zero there) and a row group
    // past the end re-reads the last row (never consumed).  A load under a branch -- lane-variant or uniform -- makes
    // the compiler wait with vmcnt(0) before the first use, which also waits for the NEXT batch just requested and
    // serialises the double buffer (measured: 5.27 -> 4.87 ms per 8-sequence step for the lane-variant guards alone).
    auto load_w = [&](u32x4 (&wq)[GM_U], int rgq, int ksq) {
        const int row = min(rgq * 4 + i4, N - 1);            // weight row whose bytes this lane loads
        const uint16_t* wp = a.W + (size_t)row * a.ldw + k0 + blk * 8;
#pragma unroll
        for (int u = 0; u < GM_U; ++u) {
            const int ko = ((ksq + u) * 128 + blk * 8 < kt) ? (ksq + u) * 128 : -blk * 8;
            wq[u] = GM_NT ? ld_nt16(wp + ko) : ld16(wp + ko);
        }
    };
    auto consume = [&](const u32x4 (&wq)[GM_U], int ks) {
#pragma unroll
        for (int u = 0; u < GM_U; ++u) {                     // steps past the slice multiply zeros (x is zero-padded to GM_KT)
            const bf16x4 w_a = __builtin_bit_cast(bf16x4, (u32x2){wq[u][0], wq[u][1]});     // k 0..3 of the chunk
            const bf16x4 w_b = __builtin_bit_cast(bf16x4, (u32x2){wq[u][2], wq[u][3]});     // k 4..7
            const int ko = (ks + u) * 128;
            const u32x4 h0 = *(const u32x4*)(xh0 + ko), l0 = *(const u32x4*)(xl0 + ko);
            c0[0] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(w_a, __builtin_bit_cast(bf16x4, (u32x2){h0[0], h0[1]}), c0[0], 0, 0, 0);
            c0[1] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(w_b, __builtin_bit_cast(bf16x4, (u32x2){h0[2], h0[3]}), c0[1], 0, 0, 0);
            c0[2] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(w_a, __builtin_bit_cast(bf16x4, (u32x2){l0[0], l0[1]}), c0[2], 0, 0, 0);
            c0[3] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(w_b, __builtin_bit_cast(bf16x4, (u32x2){l0[2], l0[3]}), c0[3], 0, 0, 0);
            if (two) {
                const u32x4 h1 = *(const u32x4*)(xh1 + ko), l1 = *(const u32x4*)(xl1 + ko);
                c1[0] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(w_a, __builtin_bit_cast(bf16x4, (u32x2){h1[0], h1[1]}), c1[0], 0, 0, 0);
                c1[1] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(w_b, __builtin_bit_cast(bf16x4, (u32x2){h1[2], h1[3]}), c1[1], 0, 0, 0);
                c1[2] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(w_a, __builtin_bit_cast(bf16x4, (u32x2){l1[0], l1[1]}), c1[2], 0, 0, 0);
                c1[3] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(w_b, __builtin_bit_cast(bf16x4, (u32x2){l1[2], l1[3]}), c1[3], 0, 0, 0);
            }
        }
    };
    // fold the 16 K-chunk blocks of a finished row group, apply the epilogue, clear the accumulators
    auto finish = [&](int rg) {
        f32x4 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc0[r] = (c0[0][r] + c0[1][r]) + (c0[2][r] + c0[3][r]);
            acc1[r] = (c1[0][r] + c1[1][r]) + (c1[2][r] + c1[3][r]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { c0[q] = (f32x4){0.f, 0.f, 0.f, 0.f}; c1[q] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        // D layout: lane 4 * blk + j holds D[i = 0..3][j] of block blk; fold the 16 blocks (K chunks)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int o = 4; o < 64; o <<= 1) {
                acc0[r] += __shfl_xor(acc0[r], o);
                if (two) acc1[r] += __shfl_xor(acc1[r], o);
            }
        }
        // lanes 0..3 (blk 0): sequence j = i4 (acc0) and 4 + i4 (acc1), weight rows rg * 4 + r
        if (blk == 0) {
            const int r0 = rg * 4;
#pragma unroll
            for (int half = 0; half < (two ? 2 : 1); ++half) {
                const int m = half * 4 + i4;
                if (m >= nseq) continue;
                const f32x4 acc = half ? acc1 : acc0;
                const float scl = half ? sc1 : sc0;
                if (EPI == EPI_SILUMUL) {                    // rows 2q = gate_q, 2q + 1 = up_q
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        if (r0 + 2 * q + 1 < N) {
                            const float gte = acc[2 * q] * scl, up = acc[2 * q + 1] * scl;
                            yg[(size_t)m * a.ldy + (r0 >> 1) + q] = (gte / (1.0f + expf(-gte))) * up;
                        }
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (r0 + r >= N) continue;
                        const float v = acc[r] * scl;
                        const size_t o = (size_t)m * a.ldy + r0 + r;
                        if (nkt > 1) atomicAdd(&yg[o], v);   // split K: partial sums onto the residual / the zeroed output
                        else if (EPI == EPI_RESADD) {
                            // in-place residual (the decoder's only use): a fire-and-forget f32 atomic is the same single
                            // add and, unlike load + store, does not make the wave drain its weight loads (vmcnt(0))
                            if (a.res == a.y) atomicAdd(&yg[o], v);
                            else yg[o] = resg[o] + v;
                        }
                        else yg[o] = v;
                        if (EPI == EPI_ARGMAX) {
                            const int ix = r0 + r + a.idx_base;
                            float& bb = half ? best1 : best0; int& bi = half ? besti1 : besti0;
                            if (v > bb || (v == bb && ix < bi)) { bb = v; bi = ix; }
                        }
                    }
                }
            }
        }
    };
    // The weight stream is ONE continuous pipeline over (row group, k-batch): two statically named register sets; the
    // next batch -- of this row group or the first of the next one -- is requested before the current one is consumed.
    int rg = bgrp * GM_W + wave, ks = 0;
    if (rg < G) load_w(wa, rg, 0);
    while (rg < G) {
        int nrg = rg, nk = ks + GM_U;
        if (nk >= nks) { nrg = rg + rg_stride; nk = 0; }
        load_w(wb, nrg, nk);
        consume(wa, ks);
        if (nk == 0) finish(rg);
        rg = nrg; ks = nk;
        if (rg >= G) break;
        nrg = rg; nk = ks + GM_U;
        if (nk >= nks) { nrg = rg + rg_stride; nk = 0; }
        load_w(wa, nrg, nk);
        consume(wb, ks);
        if (nk == 0) finish(rg);
        rg = nrg; ks = nk;
    }
    if (EPI == EPI_ARGMAX) {
        __syncthreads();
        float* rb = fsc + GM_MB; int* ri = (int*)(rb + GM_W * GM_MB);
        if (blk == 0) {
            rb[wave * GM_MB + i4] = best0; ri[wave * GM_MB + i4] = besti0;
            rb[wave * GM_MB + 4 + i4] = best1; ri[wave * GM_MB + 4 + i4] = besti1;
        }
        __syncthreads();
        if (tid < GM_MB && tid < nseq) {
            float b = rb[tid]; int bi = ri[tid];
            for (int w = 1; w < GM_W; ++w)
                if (rb[w * GM_MB + tid] > b || (rb[w * GM_MB + tid] == b && ri[w * GM_MB + tid] < bi)) { b = rb[w * GM_MB + tid]; bi = ri[w * GM_MB + tid]; }
            a.pmax[(size_t)(grp * GM_MB + tid) * gridDim.x + blockIdx.x] = b;
            a.pidx[(size_t)(grp * GM_MB + tid) * gridDim.x + blockIdx.x] = bi;
        }
        // several groups: this block's column of the OTHER groups' rows must not win (argmax_final scans every column)
        if (ngrp > 1 && tid < ngrp * GM_MB && (tid / GM_MB) != grp && tid < a.n_seq) {
            a.pmax[(size_t)tid * gridDim.x + blockIdx.x] = -INFINITY;
            a.pidx[(size_t)tid * gridDim.x + blockIdx.x] = 0x7FFFFFFF;
        }
    }
}

// usable: 3..64 sequences (9..64: 2 / 4 / 8 groups of <= 8 share the weight stream through the L2) (measured on Qwen3-8B, ms/step VALU gemvb vs this kernel: 2 seq 4.12 / 4.51, 4 seq 4.88 / 4.73,
// 8 seq 7.85 / 5.27), K % 8 == 0, and no K split for the epilogues that need complete sums
bool gemvm_ok(int epi, int n_seq, int K) {
    static int min_seq = -1;
    if (min_seq < 0) { min_seq = 3; if (const char* e = getenv("CM_GEMVM_MIN")) min_seq = atoi(e); }
    if (n_seq < min_seq || n_seq > 8 * GM_MB || K % 8 != 0) return false;
    const int nkt = (K + GM_KT - 1) / GM_KT;
    return nkt == 1 || epi == EPI_STORE || epi == EPI_RESADD;
}
int gemvm_nkt(int K) { return (K + GM_KT - 1) / GM_KT; }
int gemvm_grid(int N, int K, int num_cu, int n_seq) {
    const int nkt = gemvm_nkt(K), G = (N + 3) / 4;
    if (n_seq <= GM_MB) {
        const int per = std::max(1, std::min((G + GM_W - 1) / GM_W, std::max(1, num_cu / nkt)));
        return per * nkt;
    }
    // 2 / 4 / 8 sequence groups: 1 / ngrp of the CUs per group; logical blocks per group a multiple of 8 (the id mapping) and of nkt
    const int ngrp = n_seq > 4 * GM_MB ? 8 : (n_seq > 2 * GM_MB ? 4 : 2);
    int per = std::max(1, std::min((G + GM_W - 1) / GM_W, std::max(1, num_cu / ngrp / nkt)));
    while ((per * nkt) % 8 != 0) ++per;
    return ngrp * per * nkt;
}

template <int PRO, int EPI>
static void launch_gemvm_t(const GemvBArgs& a, int grid, hipStream_t s) {
    const size_t ldsb = (size_t)2 * GM_MB * GM_LD * 2 + (GM_MB + 2 * GM_W * GM_MB) * 4 + 64;
    static DevOnce attr;
    attr.run([] {
        (void)hipFuncSetAttribute((const void*)gemvm_kernel<PRO, EPI, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)gemvm_kernel<PRO, EPI, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    if (a.n_seq > 4) hipLaunchKernelGGL((gemvm_kernel<PRO, EPI, true>), dim3(grid), dim3(64 * GM_W), ldsb, s, a, gemvm_nkt(a.K));
    else hipLaunchKernelGGL((gemvm_kernel<PRO, EPI, false>), dim3(grid), dim3(64 * GM_W), ldsb, s, a, gemvm_nkt(a.K));
}

// nkt > 1 with EPI_STORE: the caller zeroes y first
void launch_gemvm(int pro, int epi, const GemvBArgs& a, int grid, hipStream_t s) {
    if (pro == PRO_PLAIN) {
        if (epi == EPI_STORE) launch_gemvm_t<PRO_PLAIN, EPI_STORE>(a, grid, s);
        else if (epi == EPI_RESADD) launch_gemvm_t<PRO_PLAIN, EPI_RESADD>(a, grid, s);
        else if (epi == EPI_SILUMUL) launch_gemvm_t<PRO_PLAIN, EPI_SILUMUL>(a, grid, s);
        else launch_gemvm_t<PRO_PLAIN, EPI_ARGMAX>(a, grid, s);
    } else {
        if (epi == EPI_STORE) launch_gemvm_t<PRO_RMSNORM, EPI_STORE>(a, grid, s);
        else if (epi == EPI_RESADD) launch_gemvm_t<PRO_RMSNORM, EPI_RESADD>(a, grid, s);
        else if (epi == EPI_SILUMUL) launch_gemvm_t<PRO_RMSNORM, EPI_SILUMUL>(a, grid, s);
        else launch_gemvm_t<PRO_RMSNORM, EPI_ARGMAX>(a, grid, s);
    }
}

}  // namespace cm
