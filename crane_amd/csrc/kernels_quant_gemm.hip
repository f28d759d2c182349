// Decode groups over quantised weights on the int8 matrix cores (round 4; SURVEY 8f rank 3, the review's "int8-MFMA GEMM for
// quantised decode groups").  The batched integer-dot GEMV (gemvqb_i8_kernel, kernels_quant.hip) decodes a weight row once per
// <= 8 sequences on the VALU (v_dot4): a 64-sequence round of Qwen3-8B in Q8_0 ran 2.6 K tok/s against 7.4 K on bf16 weights.
// Here a group of up to 128 sequences is ONE pass over the codes:
//
//   quant_rows_q8_kernel : every activation row -> Q8_0 blocks (32 codes + an f16-rounded scale), the arithmetic of the GEMV
//                          kernels' prologue bit for bit (RMSNorm sums in the same order, d = amax / 127, roundf(x / d)) --
//                          once per projection input, not once per workgroup;
//   gemm_q8_i8_kernel    : C[m][n] = sum over 32-element blocks b of (dw[n][b] * dx[m][b]) * (int32 dot of the block's codes) --
//                          ggml_vec_dot_q8_0_q8_0 per (m, n): the integer dot of a block is ONE v_mfma_i32_32x32x32_i8 (a lane's 16
//                          bytes of a row are k 16 h .. 16 h + 15 of the block, weights and activations alike), the block scales are
//                          applied on the VALU to the 16 int32 a lane gets per (m-tile, block).  Weight AND activation codes of a group
//                          of 4 or 8 blocks reach the workgroup as whole 128- / 256-byte row segments (coalesced, one group ahead in
//                          registers) and go through double-buffered LDS panels, where the waves pick their MFMA fragments (round 5;
//                          round 4 loaded the weight fragments straight from memory, 32 rows x 32 bytes per instruction: every cache
//                          line touched by four instructions -- that, not the arithmetic, bound the kernel).
//                          Split K over workgroups to fill the chip; f32 partials [ks][M][N]; an unsplit launch stores (and applies
//                          SiLU(gate) * up, and quantises those rows for the down projection) itself;
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
                                                            signed char* __restrict__ xq, float* __restrict__ xd, int K, int xs) {
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
            if ((t2 & 7) == 0) xd[(size_t)(k4 >> 3) * xs + m] = f16r(d);
        }
    }
}

void launch_quant_rows_q8(const float* x, int ldx, const float* nw, float eps, signed char* xq, float* xd, int M, int K, hipStream_t s, int xs) {
    if (nw) hipLaunchKernelGGL(quant_rows_q8_kernel<true>, dim3(M), dim3(256), 0, s, x, ldx, nw, eps, xq, xd, K, xs);
    else hipLaunchKernelGGL(quant_rows_q8_kernel<false>, dim3(M), dim3(256), 0, s, x, ldx, nw, eps, xq, xd, K, xs);
}

// Workgroup tile: 128 weight rows x all M activation rows; 4 MH waves = 4 n-strips of 32 weight rows x MH halves of the activation
// rows; a wave multiplies its strip with MT m-tiles of 32 rows (M <= 32 MT MH).  MH = 2 (M > 64): 512 threads, ONE workgroup per CU
// at 2 waves per SIMD -- 256 workgroups fill the chip, half the K split (and half the f32 partials) of 4-wave workgroups.
// K is walked in GROUPS of QG = 8 blocks (256 codes per row):
//   * weights: a lane's 8 x 16 bytes of the NEXT group and its 8 block scales (one 16-byte load) are requested while the current
//     group is multiplied -- 8 KB per wave one group (~2 us of work) ahead (the two halves of a strip request the same bytes: the
//     second is served by the CU's vector cache);
//   * activation codes: the group's panel (M rows x 256 bytes) is fetched ONCE per workgroup (into registers one group ahead, then
//     into one of two LDS panels, the group's block scales next to it), the waves read their fragments from there;
//   * orientation: A = weights, B = activations, so a lane's 16 results of an MFMA are 16 WEIGHT rows x ONE activation row (m = lane
//     % 32): the activation scale is one f32 per lane and step (one 4-byte LDS read), the 16 weight scales are the same for every
//     m-tile of the block (4 x 16-byte reads per BLOCK from a wave-private LDS copy of the strip's scales).  With the operands the
//     other way round (round 4) every (m-tile, block) step read 16 activation scales: 5 x ds_read_b128 per step and wave = as many
//     LDS cycles per CU as the scaling has VALU cycles;
//   * the steps of a group are software-pipelined inside the wave: fragment + scale reads two steps ahead, the MFMA one step
//     ahead of the scaling that consumes it.  Step by step (round 4: read -> MFMA -> convert -> scale, fenced per step so that the
//     register allocator survives) one wave took ~540 cycles per step for ~140 cycles of VALU work, and a second wave per SIMD did
//     not overlap it (measured with the K split forced to 1: 7.2 us per group and workgroup);
//   * scaling: 16 int32 per lane and (m-tile, block) -> f32, times dw[n] * dx[m], as packed f32 math (v_pk_mul / v_pk_fma):
//     32 VALU instructions per step -- the floor of this kernel (39 us per Qwen3-8B layer at 128 rows), not the matrix cores.
// The K split is any ks <= 16, slice i = groups [i G / ks, (i + 1) G / ks).
constexpr int QG_MIN = 8;                           // K % (32 QG_MIN) == 0: every geometry's group divides the row

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// Round 6: prompt passes (M > 128) -- geometry <2, 4, 4>: 8 waves, 256 activation rows per workgroup (a weight panel in LDS is multiplied
// with 4 m-tiles per wave: half the weight fetches per row of the 128-row geometries), and a launch covers ALL rows: blockIdx also
// walks the m-panels (a.mpan), the block scales of the rows sit a.xs floats apart ([K / 32][xs], xs >= the launch's rows).
template <int MH, int MT, int QG>
__global__ __launch_bounds__(256 * MH, (MH * MT >= 8 ? 1 : 2)) void gemm_q8_i8_kernel(QGemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char qlds[];
    constexpr int QROWB = QG * 32 + 16;                 // bytes of a panel row in LDS (padded: the 16 rows of a fragment read start 4 banks apart)
    constexpr int RC = 2 * QG;                          // 16-byte chunks of a row and group
    constexpr int NT = 256 * MH, PR = 32 * MT * MH, NCH = PR * RC / NT, NWC = 128 * RC / NT, NS = QG * MT;
    constexpr int PANEL = PR * QROWB, WPANEL = 128 * QROWB;
    constexpr int XM = PR > QGEMM_MAXM ? PR : QGEMM_MAXM;   // rows of the activation-scale panel in LDS
    typedef uint32_t scl_t __attribute__((ext_vector_type(QG / 2)));       // a row's QG f16 block scales of a group
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5, ns = wave & 3, mh = wave >> 2;
    const int K = a.w.K, N = a.w.N, nkb_all = K >> 5, G = nkb_all / QG;
    float* xds = (float*)qlds;                                              // [2][QG][XM] block scales of the activation rows, per group
    float* dwl = xds + 2 * QG * XM + wave * (QG * 32);                      // [QG][32] this wave's weight scales of the group (wave-private)
    unsigned char* Ws = qlds + (size_t)(2 * QG * XM + 4 * MH * QG * 32) * sizeof(float);   // [2][128][QROWB] weight codes of a group
    unsigned char* As = Ws + 2 * WPANEL;                                    // [2][PR][QROWB] activation codes of a group
    const int tiles = N / 128;
    // block -> (K slice, weight tile, m-panel).  Several m-panels (a prompt pass): workgroups are dealt round-robin to the 8 XCDs, so
    // the panels of one weight tile get ids 8 apart -- they run on ONE XCD, back to back, and share the tile through that L2
    const int per = tiles * a.mpan;
    const int ks = (int)blockIdx.x / per, rem = (int)blockIdx.x % per;
    int tn, mp;
    if (a.mpan > 1 && (tiles & 7) == 0) { const int xcd = rem & 7, idx = rem >> 3; tn = (idx / a.mpan) * 8 + xcd; mp = idx % a.mpan; }
    else { tn = rem % tiles; mp = rem / tiles; }
    const int m_base = mp * PR;
    const int g0 = ks * G / a.ksplit, ngrp = (ks + 1) * G / a.ksplit - g0, kb0 = g0 * QG;
    // weight scales: lane (r, .) of strip ns owns row tn * 128 + ns * 32 + r (8 f16 = one 16-byte load per group)
    const uint16_t* dp = (const uint16_t*)a.w.p1 + (size_t)(tn * 128 + ns * 32 + r) * nkb_all + kb0;
    // code panels: chunk c = tid + NT i: row c / 16, 16-byte chunk c % 16 of the group's 256 bytes -- 16 lanes fetch one row's 256
    // contiguous bytes (two whole cache lines; as MFMA fragments straight from memory a wave's load touched 32 rows x 32 bytes, every
    // line four times by four different instructions: the kernel was bound by exactly that, not by its arithmetic)
    const uint8_t* wsrc[NWC];
    const signed char* xsrc[NCH];
    int wdst[NWC], xdst[NCH];
#pragma unroll
    for (int i = 0; i < NWC; ++i) {
        const int c = tid + NT * i, row = c / RC, q = c % RC;
        wsrc[i] = a.w.p0 + (size_t)(tn * 128 + row) * K + (size_t)kb0 * 32 + 16 * q;
        wdst[i] = row * QROWB + 16 * q;
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = tid + NT * i, row = c / RC, q = c % RC;
        xsrc[i] = a.xq + (size_t)min(m_base + row, a.M - 1) * K + (size_t)kb0 * 32 + 16 * q;       // (rows past M: clamped, dropped at the store)
        xdst[i] = row * QROWB + 16 * q;
    }
    // Round 6: the weight requests run PFW groups ahead (a ring of PFW register sets, the group loop unrolled PFW times so that the sets
    // are statically named).  With ONE group ahead a workgroup had 16 KB of weights in flight: 256 CUs x 16 KB per ~2 us of a loaded HBM
    // round trip = 2 TB/s -- what the 128-row decode groups measured (1.56 TB/s, SQ_WAIT_ANY 32-41 % of the wave cycles); the activation
    // panel (L2-resident) stays one group ahead.  Requests past the slice's last group re-read it (unconditional loads).
    constexpr int PFW = (QG == 8 || MH == 1) ? 1 : 2;   // (groups of 8 blocks are 32 KB each; the 4-wave geometries run two workgroups per CU and measured 3-8 % slower with the ring: one ahead)
    u32x4 wreg[NWC], areg[NCH], wring[PFW][NWC];
    scl_t scn, sring[PFW];
    auto load_w = [&](int g, u32x4 (&w)[NWC], scl_t& sc) __attribute__((always_inline)) {
        const int gg = min(g, ngrp - 1);
#pragma unroll
        for (int i = 0; i < NWC; ++i) w[i] = ld_nt16(wsrc[i] + (size_t)gg * QG * 32);
        sc = *(const scl_t*)(dp + gg * QG);
    };
#pragma unroll
    for (int i = 0; i < NWC; ++i) wreg[i] = ld_nt16(wsrc[i]);
    scn = *(const scl_t*)dp;
#pragma unroll
    for (int u = 0; u < PFW; ++u) load_w(1 + u, wring[u], sring[u]);
#pragma unroll
    for (int i = 0; i < NCH; ++i) areg[i] = *(const u32x4*)xsrc[i];
    // (a group's scales: QG x 128 floats = one 16-byte load per thread of the first QG / 2 waves)
    constexpr int XT = QG * XM / 4, XR = XM / 4;
    const float* xdsrc = a.xd + ((size_t)kb0 + (tid % XT) / XR) * a.xs + m_base + 4 * ((tid % XT) % XR);      // block (tid / XR) of the group, rows 4 (tid % XR) ...
    const size_t xdstep = (size_t)QG * a.xs;
    f32x4 xreg = {0.f, 0.f, 0.f, 0.f};
    if (tid < XT) { xreg = *(const f32x4*)xdsrc; ((f32x4*)xds)[tid] = xreg; }
#pragma unroll
    for (int i = 0; i < NWC; ++i) *(u32x4*)(Ws + wdst[i]) = wreg[i];
#pragma unroll
    for (int i = 0; i < NCH; ++i) *(u32x4*)(As + xdst[i]) = areg[i];
    // the strip's weight scales of a group, f32, where every lane of the wave can read them: lane (r, h) writes blocks (QG / 2) h ... of
    // row r (wave-private: the wave's own DS operations are ordered, no barrier)
    auto put_dw = [&](const scl_t& sv8) {
        float* d = dwl + (QG / 2 * h) * 32 + r;
#pragma unroll
        for (int e = 0; e < QG / 4; ++e) {
            const uint32_t p = h ? sv8[QG / 4 + e] : sv8[e];
            d[64 * e] = f16bits(p & 0xFFFFu); d[64 * e + 32] = f16bits(p >> 16);
        }
    };
    put_dw(scn);
    f32x2 acc[MT][8];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[mt][i] = (f32x2){0.f, 0.f};
    __syncthreads();
    const int wrow = (ns * 32 + r) * QROWB + 16 * h;                         // + j * 32
    const int arow = (mh * MT * 32 + r) * QROWB + 16 * h;                    // + mt * 32 * QROWB + j * 32
    const int xrow = mh * MT * 32 + r;                                       // + j * XM + mt * 32
    // one group: multiply group g out of LDS buffer g & 1; then the panels of group g + 1 (weights: the ring set `wn`, requested PFW groups
    // ago) go to the other buffer and `wn` is requested again for group g + 1 + PFW
    auto group = [&](int g, u32x4 (&wn)[NWC], scl_t& scnx) __attribute__((always_inline)) {
        const bool more = g + 1 < ngrp;
        if (more) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) areg[i] = *(const u32x4*)(xsrc[i] + (size_t)(g + 1) * QG * 32);
            if (tid < XT) xreg = *(const f32x4*)(xdsrc + (size_t)(g + 1) * xdstep);
        }
        const unsigned char* Wp = Ws + (g & 1) * WPANEL + wrow;
        const unsigned char* Ap = As + (g & 1) * PANEL + arow;
        const float* xg = xds + (g & 1) * (QG * XM) + xrow;
        u32x4 avq[2], wfq[2];
        float dxq[3];
        i32x16 cq[2];
        f32x4 dwq[4];
        const i32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        // step s = (block j = s / MT, m-tile mt = s % MT)
#define Q8_LOAD(s_, slot_) do { avq[slot_] = *(const u32x4*)(Ap + ((s_) % MT) * 32 * QROWB + ((s_) / MT) * 32); \
                                dxq[(s_) % 3] = xg[((s_) / MT) * XM + ((s_) % MT) * 32]; } while (0)
#define Q8_MFMA(s_, slot_) cq[slot_] = __builtin_amdgcn_mfma_i32_32x32x32_i8( \
            (i32x4){(int)wfq[((s_) / MT) & 1][0], (int)wfq[((s_) / MT) & 1][1], (int)wfq[((s_) / MT) & 1][2], (int)wfq[((s_) / MT) & 1][3]}, \
            (i32x4){(int)avq[slot_][0], (int)avq[slot_][1], (int)avq[slot_][2], (int)avq[slot_][3]}, zero, 0, 0, 0)
        wfq[0] = *(const u32x4*)Wp;
#pragma unroll
        for (int q = 0; q < 4; ++q) dwq[q] = *(const f32x4*)(dwl + 8 * q + 4 * h);
        Q8_LOAD(0, 0);
        Q8_LOAD(1, 1);
        if (MT == 1) wfq[1] = *(const u32x4*)(Wp + 32);
        Q8_MFMA(0, 0);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int sl = s & 1, mt = s % MT, j = s / MT;
            // the next block's weight fragment: needed by the MFMA of step (j + 1) MT, issued at step (j + 1) MT - 1
            if (MT > 1 && mt == 0 && j + 1 < QG) wfq[(j + 1) & 1] = *(const u32x4*)(Wp + (j + 1) * 32);
            if (s + 1 < NS) Q8_MFMA(s + 1, sl ^ 1);
            if (MT == 1 && j + 2 < QG) wfq[j & 1] = *(const u32x4*)(Wp + (j + 2) * 32);
            if (s + 2 < NS) Q8_LOAD(s + 2, sl);
            __builtin_amdgcn_sched_barrier(0);                              // (the MFMA in front of the scaling it runs under, not behind it)
            const float dx = dxq[s % 3];
            f32x2 sv[8];
#pragma unroll
            for (int p = 0; p < 8; ++p) sv[p] = (f32x2){dwq[p >> 1][2 * (p & 1)], dwq[p >> 1][2 * (p & 1) + 1]} * (f32x2){dx, dx};
            if (mt == MT - 1 && j + 1 < QG) {
                // the next block's weight scales: requested as soon as this block's last products are formed
#pragma unroll
                for (int p = 0; p < 8; ++p) asm volatile("" : "+v"(sv[p]));
#pragma unroll
                for (int q = 0; q < 4; ++q) dwq[q] = *(const f32x4*)(dwl + (j + 1) * 32 + 8 * q + 4 * h);
            }
            // D[i] of lane (r, h): weight row 8 (i / 4) + 4 h + i % 4 of the strip, activation row r of the m-tile
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const f32x2 cf = (f32x2){(float)cq[sl][2 * p], (float)cq[sl][2 * p + 1]};
                acc[mt][p] = __builtin_elementwise_fma(sv[p], cf, acc[mt][p]);
            }
#pragma unroll
            for (int p = 0; p < 8; ++p) asm volatile("" : "+v"(acc[mt][p]));
            __builtin_amdgcn_sched_barrier(0);
        }
#undef Q8_LOAD
#undef Q8_MFMA
        if (more) {
            unsigned char* Wn = Ws + ((g + 1) & 1) * WPANEL;
            unsigned char* An = As + ((g + 1) & 1) * PANEL;
#pragma unroll
            for (int i = 0; i < NWC; ++i) *(u32x4*)(Wn + wdst[i]) = wn[i];
#pragma unroll
            for (int i = 0; i < NCH; ++i) *(u32x4*)(An + xdst[i]) = areg[i];
            if (tid < XT) ((f32x4*)(xds + ((g + 1) & 1) * (QG * XM)))[tid] = xreg;
            put_dw(scnx);                                                   // (this group's reads of the scales are behind us)
        }
        load_w(g + 1 + PFW, wn, scnx);
        __syncthreads();
    };
    for (int g = 0; g < ngrp; g += PFW) {
#pragma unroll
        for (int u = 0; u < PFW; ++u)
            if (g + u < ngrp) group(g + u, wring[u], sring[u]);
    }
    float* P = a.ws + (size_t)ks * a.slice;
    const int nq = tn * 128 + ns * 32 + 4 * h;
    constexpr int TS = 68;                                                  // floats per row of the SiLU tile (64 outputs + pad)
    float* T = (float*)Ws;                                                  // (the last group's barrier is behind every panel read)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int ml = (mh * MT + mt) * 32 + r, m = m_base + ml;
        if (a.silu) {
            // unsplit gate|up: columns (2 k, 2 k + 1) = (gate_k, up_k) sit in one lane -- the reduction kernel's expression, no f32 round trip
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float g0 = acc[mt][2 * q][0], u0 = acc[mt][2 * q][1], g1 = acc[mt][2 * q + 1][0], u1 = acc[mt][2 * q + 1][1];
                const f32x2 o = (f32x2){(g0 / (1.0f + expf(-g0))) * u0, (g1 / (1.0f + expf(-g1))) * u1};
                // (through the LDS tile: a lane's 8-byte pieces of 32 different rows were 64 partial-line writes per instruction -- the
                // unsplit gate|up launch of a 1024-row prompt pass took 441 us against 3 x 59 for the same work per workgroup in qkv)
                *(f32x2*)(T + ml * TS + ns * 16 + 2 * h + 4 * q) = o;
            }
        } else if (m < a.M) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *(f32x4*)(P + (size_t)m * a.ldp + nq + 8 * q) = (f32x4){acc[mt][2 * q][0], acc[mt][2 * q][1], acc[mt][2 * q + 1][0], acc[mt][2 * q + 1][1]};
        }
    }
    if (a.silu) {
        // rows of the tile as whole 256-byte runs: 16 lanes x 16 bytes per row
        __syncthreads();
        for (int it = tid; it < PR * 16; it += NT) {
            const int row = it >> 4, c4 = it & 15;
            if (m_base + row < a.M) *(f32x4*)(P + (size_t)(m_base + row) * a.ldp + tn * 64 + 4 * c4) = *(const f32x4*)(T + row * TS + 4 * c4);
        }
    }
    if (a.silu && a.nxq) {
        // the tile's 64 outputs of every row are two Q8_0 blocks of the down projection's input: quantised here, 8 lanes x 4 values per
        // block -- quant_rows_q8_kernel's arithmetic (amax tree, d = amax / 127, roundf(x / d), f16-rounded scale)
        const int nkout = N >> 1, l8 = tid & 7;
        for (int it = tid >> 3; it < PR * 2; it += NT >> 3) {
            const int row = it >> 1, b = it & 1;
            const f32x4 xv = *(const f32x4*)(T + row * TS + b * 32 + 4 * l8);
            float am = fmaxf(fmaxf(fabsf(xv[0]), fabsf(xv[1])), fmaxf(fabsf(xv[2]), fabsf(xv[3])));
            am = fmaxf(am, __shfl_xor(am, 1)); am = fmaxf(am, __shfl_xor(am, 2)); am = fmaxf(am, __shfl_xor(am, 4));
            const float d = am / 127.0f;
            const float id = d != 0.f ? 1.0f / d : 0.f;
            uint32_t pk = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk |= ((uint32_t)(int)roundf(xv[e] * id) & 0xFFu) << (8 * e);
            if (m_base + row < a.M) {
                ((uint32_t*)(a.nxq + (size_t)(m_base + row) * nkout))[tn * 16 + b * 8 + l8] = pk;
                if (l8 == 0) a.nxd[(size_t)(tn * 2 + b) * a.xs + m_base + row] = f16r(d);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Round 6: Q4_K weights on the same matrix-core path (the review's "K-quant tiles on the int8 GEMM"; ggml_vec_dot_q4_K_q8_K per
// output: the activation row as Q8_K blocks -- ONE f32 scale per 256 elements, int8 codes, sums of 32 codes -- and
//   sumf += d_x d * sum_j sc_j (q4_j . q8_j)  -  d_x dmin * sum_j m_j bsum_j            over the eight 32-element sub-blocks j).
// Both terms are written as Q8_0-shaped blocks so that the loop of gemm_q8_i8_kernel carries over unchanged:
//   * sub-block j is a block of 32 int8 codes (the 4-bit codes 0 .. 15, expanded from the packed nibbles when the weight panel goes to
//     LDS: the weights stay 4.5 bits in HBM) with the f32 scale d * sc_j (exact: f16 x 6-bit) on the weight side and d_x on the
//     activation side;
//   * the min term is two VIRTUAL blocks per 256: weight codes (m_0 .. m_7, 0 ...), weight scale -dmin, against the base-128 digits of
//     the code sums -- bsum_j = 128 bh_j + bl_j, bl in [0, 127], bh in [-32, 31], both int8 -- with activation scales d_x (bl) and
//     128 d_x (bh).  One v_mfma_i32_32x32x32_i8 each: the matrix cores have the room (they were 10 % busy), the VALU does not.
// A K GROUP is half a 256-block: 4 real blocks + 1 virtual block (bl in the first half, bh in the second) = 5 steps per m-tile;
// activation rows arrive as groups of 160 bytes [128 codes | 8 digits + 24 zeros] from quant_rows_q8k_kernel, their scales as 5 f32 per
// group.  Sums differ from ggml's by the f32 rounding of (d sc_j) d_x per sub-block instead of one product per 256 (1e-7).
// Q6_K (sub-blocks of 16 codes: two scales per 32-deep MFMA step) is not on this path: such tensors keep the batched GEMV.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int QKB = 5;                               // MFMA steps per m-tile and group: 4 sub-blocks + 1 virtual block
constexpr int QKROW = 160;                           // bytes of an activation row per group

// one 256-thread block per row; one wave per 256-element block (quantize_row_q8_K as gemvq_i8_kernel's prologue computes it: the signed
// value of the FIRST element with the largest |x| -> iscale = -128 / max, q = min(127, rint(x iscale)), d = 1 / iscale)
template <bool NORM>
__global__ __launch_bounds__(256) void quant_rows_q8k_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ nw, float eps,
                                                             signed char* __restrict__ xq, float* __restrict__ xd, int K, int xs) {
    __shared__ float red[4];
    const int m = blockIdx.x, t2 = threadIdx.x, lane = t2 & 63, wave = t2 >> 6;
    const float* xr = x + (size_t)m * ldx;
    const int n4 = K >> 2;
    float r = 1.f;
    if (NORM) {                                      // (quant_rows_q8_kernel's order of the sum of squares)
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
        if (lane == 0) red[wave] = t;
        __syncthreads();
        r = 1.0f / sqrtf(((red[0] + red[1]) + (red[2] + red[3])) / (float)K + eps);
    }
    const int nblk = K >> 8;
    signed char* row = xq + (size_t)m * (size_t)(K >> 7) * QKROW;
    for (int blk = wave; blk < nblk; blk += 4) {
        const int k4 = blk * 64 + lane;
        f32x4 v = *(const f32x4*)(xr + (k4 << 2));
        if (NORM) {
            const f32x4 w = *(const f32x4*)(nw + (k4 << 2));
            v[0] = __fmul_rn(__fmul_rn(v[0], r), w[0]); v[1] = __fmul_rn(__fmul_rn(v[1], r), w[1]);
            v[2] = __fmul_rn(__fmul_rn(v[2], r), w[2]); v[3] = __fmul_rn(__fmul_rn(v[3], r), w[3]);
        }
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
        int s4 = 0; uint32_t pk = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int q = mx != 0.f ? min(127, __float2int_rn(v[e] * iscale)) : 0;
            s4 += q;
            pk |= ((uint32_t)q & 0xFFu) << (8 * e);
        }
        // lane l holds elements 4 l .. 4 l + 3 of the block: group (half) l / 32, byte 4 (l % 32) of its 128 code bytes
        signed char* g0 = row + (size_t)(2 * blk) * QKROW;
        ((uint32_t*)(g0 + (lane >> 5) * QKROW))[lane & 31] = pk;
        // sums of 32 codes = 8 consecutive lanes -> digits bl (first group's virtual block) and bh (second group's)
        int bs = s4 + __shfl_xor(s4, 1); bs += __shfl_xor(bs, 2); bs += __shfl_xor(bs, 4);
        const int bl = bs & 127, bh = (bs - bl) >> 7;                // bs = 128 bh + bl, bs in [-4096, 4064]
        if ((lane & 7) == 0) { g0[128 + (lane >> 3)] = (signed char)bl; g0[QKROW + 128 + (lane >> 3)] = (signed char)bh; }
        if (lane < 12) {                                              // the zeros behind the 8 digits (24 bytes per group = 6 dwords x 2 groups)
            const int gi = lane / 6, w = lane % 6;
            ((uint32_t*)(g0 + gi * QKROW + 136))[w] = 0u;
        }
        if (lane < 10) {                                              // scales: groups 2 blk, 2 blk + 1 x 5 blocks
            const int gi = lane / 5, j = lane % 5;
            const float d = mx != 0.f ? 1.0f / iscale : 0.f;
            xd[(size_t)((2 * blk + gi) * QKB + j) * xs + m] = (j == 4 && gi == 1) ? 128.0f * d : d;
        }
    }
}

void launch_quant_rows_q8k(const float* x, int ldx, const float* nw, float eps, signed char* xq, float* xd, int M, int K, hipStream_t s, int xs) {
    if (nw) hipLaunchKernelGGL(quant_rows_q8k_kernel<true>, dim3(M), dim3(256), 0, s, x, ldx, nw, eps, xq, xd, K, xs);
    else hipLaunchKernelGGL(quant_rows_q8k_kernel<false>, dim3(M), dim3(256), 0, s, x, ldx, nw, eps, xq, xd, K, xs);
}

// gemm_q8_i8_kernel's tiling over Q4_K weights, with the format's own lever: the 6-bit sub-scales are INTEGERS, so they are multiplied
// into the weight codes when the panel is built -- sc_j = 8 sh_j + sl_j, and q4 * sl_j, q4 * sh_j <= 15 * 7 both fit an int8 -- and the
// matrix core then sums sc_j (q4_j . q8_j) over the four sub-blocks of a group BY ITSELF, chaining its int32 accumulator through two
// planes (lo, hi) of four 32-deep steps each; what is left for the VALU per group and m-tile is t = I_lo + 8 I_hi (16 shift-adds), ONE
// float pass acc += (d d_x) t, and the virtual (min-term) block's pass: 80 VALU instructions for 9 MFMAs, where the first version of
// this kernel (every sub-block scaled in float like a Q8_0 block) needed 160 for 5 and spilled when it kept integer sums in registers.
// Weight rows arrive as 64 packed bytes per group (4 x 16-byte chunks: one per thread at 512 threads) with the row's 16-byte block
// header; a thread expands its 32 codes into both planes with two packed 16-bit multiplies per dword (no byte overflows: <= 105).
// LDS panel row: [128 lo | 128 hi | 32 virtual (m_0 .. m_7, 0 ...) | 16 pad] = 304 bytes (fragment rows 12 banks apart: conflict-free).
template <int MH, int MT>
__global__ __launch_bounds__(256 * MH, 1) void gemm_q4k_i8_kernel(QGemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char qlds[];
    constexpr int WROWB = 304, AROWB = QKROW + 16;
    constexpr int NT = 256 * MH, PR = 32 * MT * MH, NS = QKB * MT;
    constexpr int NCHT = PR * 10, NCH = (NCHT + NT - 1) / NT;          // 16-byte chunks of the activation panel of a group, per thread
    constexpr int NWC = 128 * 4 / NT;                   // packed weight chunks per thread (1 at 512 threads, 2 at 256)
    constexpr int PANEL = PR * AROWB, WPANEL = 128 * WROWB;
    constexpr int XM = QGEMM_MAXM;
    static_assert(PR <= QGEMM_MAXM, "m-panels of this kernel are at most 128 rows");
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5, ns = wave & 3, mh = wave >> 2;
    const int K = a.w.K, N = a.w.N, nb256 = K >> 8, G = K >> 7;
    float* xds = (float*)qlds;                                              // [2][QKB][XM] activation block scales of a group
    float* dwl = xds + 2 * QKB * XM + wave * 64;                            // [2][32] this wave's d and -dmin of the group's 256-block (wave-private)
    unsigned char* Ws = qlds + (size_t)(2 * QKB * XM + 4 * MH * 64) * sizeof(float);   // [2][128][WROWB]
    unsigned char* As = Ws + 2 * WPANEL;                                    // [2][PR][AROWB]
    const int tiles = N / 128;
    const int per = tiles * a.mpan;
    const int ks = (int)blockIdx.x / per, rem = (int)blockIdx.x % per;
    int tn, mp;
    if (a.mpan > 1 && (tiles & 7) == 0) { const int xcd = rem & 7, idx = rem >> 3; tn = (idx / a.mpan) * 8 + xcd; mp = idx % a.mpan; }
    else { tn = rem % tiles; mp = rem / tiles; }
    const int m_base = mp * PR;
    const int g0 = ks * G / a.ksplit, ngrp = (ks + 1) * G / a.ksplit - g0;
    const uint8_t* hdr = a.w.p1 + (size_t)(tn * 128 + ns * 32 + r) * nb256 * 16;       // lane (r, .) of strip ns owns row tn * 128 + ns * 32 + r
    // packed weight chunks: c = tid + NT i: row c / 4, chunk q = c % 4 of the group's 64 bytes: run q / 2 (sub-block 2 (q / 2) in the low
    // nibbles, + 1 in the high ones), bytes 16 (q % 2) ... of the run
    const uint8_t* wsrc[NWC];
    int wdst[NWC];
    const uint8_t* whdr[NWC];
#pragma unroll
    for (int i = 0; i < NWC; ++i) {
        const int c = tid + NT * i, row = c >> 2, q = c & 3;
        wsrc[i] = a.w.p0 + (size_t)(tn * 128 + row) * (K >> 1) + (size_t)g0 * 64 + 16 * q;
        wdst[i] = row * WROWB + (q >> 1) * 64 + (q & 1) * 16;
        whdr[i] = a.w.p1 + (size_t)(tn * 128 + row) * nb256 * 16;
    }
    const signed char* xbase = a.xq + (size_t)g0 * QKROW;
    const size_t xrow_b = (size_t)G * QKROW;
    auto xsrc_of = [&](int c) -> const signed char* {
        const int row = c / 10, q = c % 10;
        return xbase + (size_t)min(m_base + row, a.M - 1) * xrow_b + 16 * q;
    };
    // the packed weights run PFW groups ahead (8 KB per group and workgroup: with one group ahead the 128-row launches were bound by
    // the latency of that one request -- SQ_WAIT_ANY 41 % of the wave cycles at 14.5 VALU per MFMA); headers (L2 hits, shared by the two
    // groups of a 256-block and by four threads) and the activation panel one group ahead.  Ring sets are statically named: the group
    // loop is unrolled PFW times.  Requests past the slice's end re-read its last group.
    constexpr int PFW = NWC == 1 ? 4 : 2;
    u32x4 wring[PFW][NWC], hring[2][NWC], aring[2][NCH];      // (headers and activations: two groups ahead, slots by group parity)
    auto load_w = [&](int g, u32x4 (&w)[NWC]) __attribute__((always_inline)) {
        const int gg = min(g, ngrp - 1);
#pragma unroll
        for (int i = 0; i < NWC; ++i) w[i] = ld_nt16(wsrc[i] + (size_t)gg * 64);
    };
    auto load_group = [&](int g, u32x4 (&hreg)[NWC], u32x4 (&areg)[NCH]) __attribute__((always_inline)) {      // headers + activations of group g (clamped)
        const int gg = min(g, ngrp - 1);
#pragma unroll
        for (int i = 0; i < NWC; ++i) hreg[i] = *(const u32x4*)(whdr[i] + (size_t)((g0 + gg) >> 1) * 16);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = min(tid + NT * i, NCHT - 1);
            areg[i] = *(const u32x4*)(xsrc_of(c) + (size_t)gg * QKROW);
        }
    };
    // 4 codes of a dword times a factor <= 7: two 16-bit lanes, no byte overflows
    auto mul4 = [](uint32_t c, int f) -> uint32_t {
        const u16x2 v = __builtin_bit_cast(u16x2, c) * (u16x2){(unsigned short)f, (unsigned short)f};
        return __builtin_bit_cast(uint32_t, v);
    };
    auto store_group = [&](int buf, int g, const u32x4 (&wreg)[NWC], const u32x4 (&hreg)[NWC], const u32x4 (&areg)[NCH]) __attribute__((always_inline)) {
        unsigned char* Wn = Ws + buf * WPANEL;
        unsigned char* An = As + buf * PANEL;
        const int half = (g0 + g) & 1;
#pragma unroll
        for (int i = 0; i < NWC; ++i) {
            const u32x4 w = wreg[i], hb = hreg[i];
            const int q = (tid + NT * i) & 3, ja = 4 * half + 2 * (q >> 1);          // sub-blocks ja (low nibbles), ja + 1 (high nibbles) of the 256-block
            // all eight 6-bit scales and mins of the header at once (ggml's kmask unpacking of scales[12]): sc_0 .. sc_3 | sc_4 .. sc_7 | m_0 .. m_3 | m_4 .. m_7
            const uint32_t u0 = hb[1], u1 = hb[2], u2 = hb[3];
            const uint32_t sc03 = u0 & 0x3F3F3F3Fu, sc47 = (u2 & 0x0F0F0F0Fu) | (((u0 >> 6) & 0x03030303u) << 4);
            const uint32_t m03 = u1 & 0x3F3F3F3Fu, m47 = ((u2 >> 4) & 0x0F0F0F0Fu) | (((u1 >> 6) & 0x03030303u) << 4);
            const uint32_t scw = (ja < 4 ? sc03 : sc47) >> (8 * (ja & 3));
            const int sa = (int)(scw & 0xFFu), sb = (int)((scw >> 8) & 0xFFu);
            u32x4 lo_a, hi_a, lo_b, hi_b;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t L = w[e] & 0x0F0F0F0Fu, Hn = (w[e] >> 4) & 0x0F0F0F0Fu;
                lo_a[e] = mul4(L, sa & 7); hi_a[e] = mul4(L, sa >> 3);
                lo_b[e] = mul4(Hn, sb & 7); hi_b[e] = mul4(Hn, sb >> 3);
            }
            *(u32x4*)(Wn + wdst[i]) = lo_a; *(u32x4*)(Wn + wdst[i] + 32) = lo_b;
            *(u32x4*)(Wn + wdst[i] + 128) = hi_a; *(u32x4*)(Wn + wdst[i] + 160) = hi_b;
            if (q == 0) {                                 // the virtual block of the row: (m_0 .. m_7, 0 ...)
                unsigned char* vb = Wn + ((tid + NT * i) >> 2) * WROWB + 256;
                *(u32x4*)vb = (u32x4){m03, m47, 0u, 0u};
                *(u32x4*)(vb + 16) = (u32x4){0u, 0u, 0u, 0u};
            }
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + NT * i;
            if (c < NCHT) *(u32x4*)(An + (c / 10) * AROWB + 16 * (c % 10)) = areg[i];
        }
    };
    // d and -dmin of the strip's rows for group g: lane (r, h = 0) writes d, h = 1 -dmin
    uint32_t shdr0 = 0;
    auto load_scales = [&](int g) { shdr0 = *(const uint32_t*)(hdr + (size_t)((g0 + g) >> 1) * 16); };
    auto put_dw = [&]() { dwl[h * 32 + r] = h == 0 ? f16bits(shdr0 & 0xFFFFu) : -f16bits(shdr0 >> 16); };
    constexpr int XT = QKB * XM / 4, XR = XM / 4;       // 160 float4 per group
    const bool xdo = tid < XT;
    const float* xdsrc = a.xd + ((size_t)g0 * QKB + (xdo ? tid / XR : 0)) * a.xs + m_base + 4 * ((xdo ? tid : 0) % XR);
    const size_t xdstep = (size_t)QKB * a.xs;
    f32x4 xreg = {0.f, 0.f, 0.f, 0.f};
    {
        u32x4 w0[NWC];
        load_w(0, w0);
#pragma unroll
        for (int u = 0; u < PFW; ++u) load_w(1 + u, wring[u]);
        load_group(0, hring[0], aring[0]); load_scales(0);
        if (xdo) xreg = *(const f32x4*)xdsrc;
        store_group(0, 0, w0, hring[0], aring[0]); put_dw();
        if (xdo) ((f32x4*)xds)[tid] = xreg;
        load_group(1, hring[1], aring[1]);
    }
    f32x2 acc[MT][8];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[mt][i] = (f32x2){0.f, 0.f};
    __syncthreads();
    const int wrow = (ns * 32 + r) * WROWB + 16 * h;
    const int arow = (mh * MT * 32 + r) * AROWB + 16 * h;
    const int xrow = mh * MT * 32 + r;
    auto group = [&](int g, u32x4 (&wn)[NWC], u32x4 (&hn)[NWC], u32x4 (&an)[NCH], u32x4 (&hf)[NWC], u32x4 (&af)[NCH]) __attribute__((always_inline)) {
        // hn / an: headers and activations of group g + 1 (requested one group ago); hf / af: the set that is free now -> group g + 2
        const bool more = g + 1 < ngrp;
        load_group(g + 2, hf, af);
        if (more) { load_scales(g + 1); if (xdo) xreg = *(const f32x4*)(xdsrc + (size_t)(g + 1) * xdstep); }
        const unsigned char* Wp = Ws + (g & 1) * WPANEL + wrow;
        const unsigned char* Ap = As + (g & 1) * PANEL + arow;
        const float* xg = xds + (g & 1) * (QKB * XM) + xrow;
        u32x4 avq[2], wl[2], wh[2];
        float dxq[3];
        i32x16 il[MT], ih[MT], cv[2];
        const i32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        // step s = (block j = s / MT, m-tile mt = s % MT); blocks 0 .. 3: two chained MFMAs (planes lo / hi), block 4: the virtual one
#define QK_LOAD(s_, slot_) do { avq[slot_] = *(const u32x4*)(Ap + ((s_) % MT) * 32 * AROWB + ((s_) / MT) * 32); \
                                dxq[(s_) % 3] = xg[((s_) / MT) * XM + ((s_) % MT) * 32]; } while (0)
#define QK_I4(v_) (i32x4){(int)(v_)[0], (int)(v_)[1], (int)(v_)[2], (int)(v_)[3]}
#define QK_MFMA(s_, slot_) do { constexpr int j_ = (s_) / MT, m_ = (s_) % MT; \
            if (j_ < 4) { il[m_] = __builtin_amdgcn_mfma_i32_32x32x32_i8(QK_I4(wl[j_ & 1]), QK_I4(avq[slot_]), j_ == 0 ? zero : il[m_], 0, 0, 0); \
                          ih[m_] = __builtin_amdgcn_mfma_i32_32x32x32_i8(QK_I4(wh[j_ & 1]), QK_I4(avq[slot_]), j_ == 0 ? zero : ih[m_], 0, 0, 0); } \
            else cv[slot_] = __builtin_amdgcn_mfma_i32_32x32x32_i8(QK_I4(wl[j_ & 1]), QK_I4(avq[slot_]), zero, 0, 0, 0); } while (0)
        // weight fragments of block j: lo plane at 32 j, hi plane at 128 + 32 j; the virtual block (j = 4) at 256 (into wl)
        auto wfrag = [&](int j) __attribute__((always_inline)) {
            wl[j & 1] = *(const u32x4*)(Wp + (j < 4 ? j * 32 : 256));
            if (j < 4) wh[j & 1] = *(const u32x4*)(Wp + 128 + j * 32);
        };
        wfrag(0);
        QK_LOAD(0, 0);
        QK_LOAD(1, 1);
        if (MT == 1) wfrag(1);
        QK_MFMA(0, 0);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int sl = s & 1, mt = s % MT, j = s / MT;
            if (MT > 1 && mt == 0 && j + 1 < QKB) wfrag(j + 1);
            if (s + 1 < NS) {
                // (macro needs a compile-time step: the loop is fully unrolled, `s` is a constant in every copy)
                switch (s + 1) {
#define QK_CASE(n_) case n_: if constexpr (n_ < NS) { QK_MFMA(n_, (n_) & 1); } break;
                    QK_CASE(1) QK_CASE(2) QK_CASE(3) QK_CASE(4) QK_CASE(5) QK_CASE(6) QK_CASE(7) QK_CASE(8) QK_CASE(9)
                    QK_CASE(10) QK_CASE(11) QK_CASE(12) QK_CASE(13) QK_CASE(14) QK_CASE(15) QK_CASE(16) QK_CASE(17) QK_CASE(18) QK_CASE(19)
#undef QK_CASE
                    default: break;
                }
            }
            if (MT == 1 && j + 2 < QKB) wfrag(j + 2);
            if (s + 2 < NS) QK_LOAD(s + 2, sl);
            __builtin_amdgcn_sched_barrier(0);
            const float dx = dxq[s % 3];
            if (j == 3) {
                // the group's real term for this m-tile: t = I_lo + 8 I_hi, acc += (d d_x) t
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 dd = *(const f32x4*)(dwl + 8 * q + 4 * h);
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int p = 2 * q + e;
                        const f32x2 sv = (f32x2){dd[2 * e], dd[2 * e + 1]} * (f32x2){dx, dx};
                        const f32x2 cf = (f32x2){(float)(il[mt][2 * p] + (ih[mt][2 * p] << 3)), (float)(il[mt][2 * p + 1] + (ih[mt][2 * p + 1] << 3))};
                        acc[mt][p] = __builtin_elementwise_fma(sv, cf, acc[mt][p]);
                    }
                }
            } else if (j == 4) {
                // the virtual block (min term): scale -dmin * (d_x | 128 d_x) on the digit dot product
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 dm = *(const f32x4*)(dwl + 32 + 8 * q + 4 * h);
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int p = 2 * q + e;
                        const f32x2 sv = (f32x2){dm[2 * e], dm[2 * e + 1]} * (f32x2){dx, dx};
                        const f32x2 cf = (f32x2){(float)cv[sl][2 * p], (float)cv[sl][2 * p + 1]};
                        acc[mt][p] = __builtin_elementwise_fma(sv, cf, acc[mt][p]);
                    }
                }
            }
#pragma unroll
            for (int p = 0; p < 8; ++p) asm volatile("" : "+v"(acc[mt][p]));
            __builtin_amdgcn_sched_barrier(0);
        }
#undef QK_LOAD
#undef QK_MFMA
#undef QK_I4
        if (more) { store_group((g + 1) & 1, g + 1, wn, hn, an); put_dw(); if (xdo) ((f32x4*)(xds + ((g + 1) & 1) * (QKB * XM)))[tid] = xreg; }
        load_w(g + 1 + PFW, wn);
        __syncthreads();
    };
    for (int g = 0; g < ngrp; g += PFW) {
#pragma unroll
        for (int u = 0; u < PFW; ++u)
            if (g + u < ngrp) group(g + u, wring[u], hring[(u + 1) & 1], aring[(u + 1) & 1], hring[u & 1], aring[u & 1]);
    }
    float* P = a.ws + (size_t)ks * a.slice;
    const int nq = tn * 128 + ns * 32 + 4 * h;
    constexpr int TS = 68;
    float* T = (float*)Ws;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int ml = (mh * MT + mt) * 32 + r, m = m_base + ml;
        if (a.silu) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float g0v = acc[mt][2 * q][0], u0 = acc[mt][2 * q][1], g1v = acc[mt][2 * q + 1][0], u1 = acc[mt][2 * q + 1][1];
                const f32x2 o = (f32x2){(g0v / (1.0f + expf(-g0v))) * u0, (g1v / (1.0f + expf(-g1v))) * u1};
                *(f32x2*)(T + ml * TS + ns * 16 + 2 * h + 4 * q) = o;
            }
        } else if (m < a.M) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *(f32x4*)(P + (size_t)m * a.ldp + nq + 8 * q) = (f32x4){acc[mt][2 * q][0], acc[mt][2 * q][1], acc[mt][2 * q + 1][0], acc[mt][2 * q + 1][1]};
        }
    }
    if (a.silu) {
        __syncthreads();
        for (int it = tid; it < PR * 16; it += NT) {
            const int row = it >> 4, c4 = it & 15;
            if (m_base + row < a.M) *(f32x4*)(P + (size_t)(m_base + row) * a.ldp + tn * 64 + 4 * c4) = *(const f32x4*)(T + row * TS + 4 * c4);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Q6_K weights (ggml_vec_dot_q6_K_q8_K: sumf += d_x d * sum_j sc_j ((q6_j - 32) . q8_j) over the SIXTEEN 16-element sub-blocks of a
// 256-block, sc_j int8) on the same path.  A sub-block is half of the 32-deep MFMA step, so the steps here are
// v_mfma_i32_32x32x16_i8: 8 per K group of 128 elements, fragments of 8 bytes per lane (ds_read_b64; panel rows 136 bytes apart: the
// 32 rows of a read land on 32 distinct 8-byte slots), weight scale d * sc_j (exact in f32) per step, activation scale d_x.  The 6-bit
// codes are assembled from ql / qh and re-centred (- 32) on their way into the LDS panel; the activation rows are the SAME Q8_K groups
// the Q4_K kernel reads (their digit block is not used here), so a Q4_K and a Q6_K projection of one input share one quantiser launch.
// Twice the VALU work of a Q8_0 tensor per weight (a scaling pass per 16 instead of per 32): format coverage -- what makes a Q4_K_M
// file (Q4_K + Q6_K tensors) run end to end on the matrix cores -- not the format's best.
// ---------------------------------------------------------------------------------------------------------------------------
template <int MH, int MT>
__global__ __launch_bounds__(256 * MH, (MH * MT >= 8 ? 1 : 2)) void gemm_q6k_i8_kernel(QGemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char qlds[];
    constexpr int ROWB = 136;                           // bytes of a panel row in LDS (128 codes + 8)
    constexpr int NT = 256 * MH, PR = 32 * MT * MH, NS = 8 * MT;
    constexpr int NCH = PR * 8 / NT;                    // 16-byte chunks of the activation panel of a group, per thread
    constexpr int NWC = 128 * 4 / NT;                   // weight pieces (8 values of l: 3 x 8 bytes -> 32 codes) per thread
    constexpr int PANEL = PR * ROWB, WPANEL = 128 * ROWB;
    constexpr int XM = PR > QGEMM_MAXM ? PR : QGEMM_MAXM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5, ns = wave & 3, mh = wave >> 2;
    const int K = a.w.K, N = a.w.N, nb256 = K >> 8, G = K >> 7;
    float* xds = (float*)qlds;                                              // [2][XM] the group's activation scale (one per 256-block)
    float* dwl = xds + 2 * XM + wave * (8 * 32);                            // [8][32] this wave's weight scales of the group
    unsigned char* Ws = qlds + (size_t)(2 * XM + 4 * MH * 8 * 32) * sizeof(float);
    unsigned char* As = Ws + 2 * WPANEL;
    const int tiles = N / 128;
    const int per = tiles * a.mpan;
    const int ks = (int)blockIdx.x / per, rem = (int)blockIdx.x % per;
    int tn, mp;
    if (a.mpan > 1 && (tiles & 7) == 0) { const int xcd = rem & 7, idx = rem >> 3; tn = (idx / a.mpan) * 8 + xcd; mp = idx % a.mpan; }
    else { tn = rem % tiles; mp = rem / tiles; }
    const int m_base = mp * PR;
    const int g0 = ks * G / a.ksplit, ngrp = (ks + 1) * G / a.ksplit - g0;
    // weight pieces: c = tid + NT i: row c / 4, l = 8 (c % 4) .. + 7 of the group's half-block
    const uint8_t *wql[NWC], *wqh[NWC];
    int wdst[NWC];
#pragma unroll
    for (int i = 0; i < NWC; ++i) {
        const int c = tid + NT * i, row = c >> 2, q = c & 3;
        wql[i] = a.w.p0 + (size_t)(tn * 128 + row) * (K >> 1) + (size_t)g0 * 64 + 8 * q;
        wqh[i] = a.w.p1 + (size_t)(tn * 128 + row) * (K >> 2) + (size_t)g0 * 32 + 8 * q;
        wdst[i] = row * ROWB + 8 * q;
    }
    const int8_t* sc_row = (const int8_t*)a.w.p2 + (size_t)(tn * 128 + ns * 32 + r) * (K >> 4);      // 8 int8 scales per group
    const uint16_t* d_row = (const uint16_t*)a.w.p3 + (size_t)(tn * 128 + ns * 32 + r) * nb256;
    const signed char* xsrc[NCH];
    int xdst[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = tid + NT * i, row = c >> 3, q = c & 7;
        xsrc[i] = a.xq + (size_t)min(m_base + row, a.M - 1) * (size_t)G * QKROW + (size_t)g0 * QKROW + 16 * q;
        xdst[i] = row * ROWB + 16 * q;
    }
    u32x2 wa[NWC], wb[NWC], wh[NWC];
    u32x4 areg[NCH];
    u32x2 screg; uint32_t dreg = 0;
    float xreg = 0.f;
    const float* xdp = a.xd + (size_t)g0 * QKB * a.xs + m_base + (tid % XM);      // scale 0 of a group's five = d_x
    auto load_group = [&](int g) {
#pragma unroll
        for (int i = 0; i < NWC; ++i) {
            wa[i] = *(const u32x2*)(wql[i] + (size_t)g * 64);
            wb[i] = *(const u32x2*)(wql[i] + (size_t)g * 64 + 32);
            wh[i] = *(const u32x2*)(wqh[i] + (size_t)g * 32);
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i) areg[i] = *(const u32x4*)(xsrc[i] + (size_t)g * QKROW);
        screg = *(const u32x2*)(sc_row + (size_t)(g0 + g) * 8);
        dreg = d_row[(g0 + g) >> 1];
        if (tid < XM) xreg = xdp[(size_t)g * QKB * a.xs];
    };
    auto recentre = [](uint32_t c) -> uint32_t { return ((c | 0x80808080u) - 0x20202020u) ^ 0x80808080u; };      // bytewise c - 32 (c in 0 .. 63)
    auto store_group = [&](int buf) {
        unsigned char* Wn = Ws + buf * WPANEL;
        unsigned char* An = As + buf * PANEL;
#pragma unroll
        for (int i = 0; i < NWC; ++i) {
            u32x2 c0, c1, c2, c3;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const uint32_t A = wa[i][e], B = wb[i][e], Hq = wh[i][e];
                c0[e] = recentre((A & 0x0F0F0F0Fu) | ((Hq & 0x03030303u) << 4));
                c1[e] = recentre((B & 0x0F0F0F0Fu) | (((Hq >> 2) & 0x03030303u) << 4));
                c2[e] = recentre(((A >> 4) & 0x0F0F0F0Fu) | (((Hq >> 4) & 0x03030303u) << 4));
                c3[e] = recentre(((B >> 4) & 0x0F0F0F0Fu) | (((Hq >> 6) & 0x03030303u) << 4));
            }
            *(u32x2*)(Wn + wdst[i]) = c0; *(u32x2*)(Wn + wdst[i] + 32) = c1; *(u32x2*)(Wn + wdst[i] + 64) = c2; *(u32x2*)(Wn + wdst[i] + 96) = c3;
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i) {             // (rows are 136 bytes apart: 8-byte aligned pieces)
            *(u32x2*)(An + xdst[i]) = (u32x2){areg[i][0], areg[i][1]};
            *(u32x2*)(An + xdst[i] + 8) = (u32x2){areg[i][2], areg[i][3]};
        }
        if (tid < XM) xds[buf * XM + tid] = xreg;
    };
    auto put_dw = [&]() {                               // lane (r, h) writes the scales of sub-blocks 4 h .. 4 h + 3 of the group
        const float d = f16bits(dreg);
        const uint32_t w = screg[h];
#pragma unroll
        for (int e = 0; e < 4; ++e) dwl[(4 * h + e) * 32 + r] = d * (float)(int)(signed char)((w >> (8 * e)) & 0xFFu);
    };
    load_group(0);
    store_group(0); put_dw();
    f32x2 acc[MT][8];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[mt][i] = (f32x2){0.f, 0.f};
    __syncthreads();
    const int wrow = (ns * 32 + r) * ROWB + 8 * h;
    const int arow = (mh * MT * 32 + r) * ROWB + 8 * h;
    const int xrow = mh * MT * 32 + r;
    for (int g = 0; g < ngrp; ++g) {
        const bool more = g + 1 < ngrp;
        if (more) load_group(g + 1);
        const unsigned char* Wp = Ws + (g & 1) * WPANEL + wrow;
        const unsigned char* Ap = As + (g & 1) * PANEL + arow;
        float dxm[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) dxm[mt] = xds[(g & 1) * XM + xrow + mt * 32];
        u32x2 avq[2], wfq[2];
        i32x16 cq[2];
        f32x4 dwq[4];
        const i32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define Q6_LOAD(s_, slot_) avq[slot_] = *(const u32x2*)(Ap + ((s_) % MT) * 32 * ROWB + ((s_) / MT) * 16)
#define Q6_MFMA(s_, slot_) cq[slot_] = __builtin_amdgcn_mfma_i32_32x32x16_i8( \
            __builtin_bit_cast(long, wfq[((s_) / MT) & 1]), __builtin_bit_cast(long, avq[slot_]), zero, 0, 0, 0)
        wfq[0] = *(const u32x2*)Wp;
#pragma unroll
        for (int q = 0; q < 4; ++q) dwq[q] = *(const f32x4*)(dwl + 8 * q + 4 * h);
        Q6_LOAD(0, 0);
        Q6_LOAD(1, 1);
        if (MT == 1) wfq[1] = *(const u32x2*)(Wp + 16);
        Q6_MFMA(0, 0);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int sl = s & 1, mt = s % MT, j = s / MT;
            if (MT > 1 && mt == 0 && j + 1 < 8) wfq[(j + 1) & 1] = *(const u32x2*)(Wp + (j + 1) * 16);
            if (s + 1 < NS) Q6_MFMA(s + 1, sl ^ 1);
            if (MT == 1 && j + 2 < 8) wfq[j & 1] = *(const u32x2*)(Wp + (j + 2) * 16);
            if (s + 2 < NS) Q6_LOAD(s + 2, sl);
            __builtin_amdgcn_sched_barrier(0);
            const float dx = dxm[mt];
            f32x2 sv[8];
#pragma unroll
            for (int p = 0; p < 8; ++p) sv[p] = (f32x2){dwq[p >> 1][2 * (p & 1)], dwq[p >> 1][2 * (p & 1) + 1]} * (f32x2){dx, dx};
            if (mt == MT - 1 && j + 1 < 8) {
#pragma unroll
                for (int p = 0; p < 8; ++p) asm volatile("" : "+v"(sv[p]));
#pragma unroll
                for (int q = 0; q < 4; ++q) dwq[q] = *(const f32x4*)(dwl + (j + 1) * 32 + 8 * q + 4 * h);
            }
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const f32x2 cf = (f32x2){(float)cq[sl][2 * p], (float)cq[sl][2 * p + 1]};
                acc[mt][p] = __builtin_elementwise_fma(sv[p], cf, acc[mt][p]);
            }
#pragma unroll
            for (int p = 0; p < 8; ++p) asm volatile("" : "+v"(acc[mt][p]));
            __builtin_amdgcn_sched_barrier(0);
        }
#undef Q6_LOAD
#undef Q6_MFMA
        if (more) { store_group((g + 1) & 1); put_dw(); }
        __syncthreads();
    }
    float* P = a.ws + (size_t)ks * a.slice;
    const int nq = tn * 128 + ns * 32 + 4 * h;
    constexpr int TS = 68;
    float* T = (float*)Ws;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int ml = (mh * MT + mt) * 32 + r, m = m_base + ml;
        if (a.silu) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float g0v = acc[mt][2 * q][0], u0 = acc[mt][2 * q][1], g1v = acc[mt][2 * q + 1][0], u1 = acc[mt][2 * q + 1][1];
                const f32x2 o = (f32x2){(g0v / (1.0f + expf(-g0v))) * u0, (g1v / (1.0f + expf(-g1v))) * u1};
                *(f32x2*)(T + ml * TS + ns * 16 + 2 * h + 4 * q) = o;
            }
        } else if (m < a.M) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *(f32x4*)(P + (size_t)m * a.ldp + nq + 8 * q) = (f32x4){acc[mt][2 * q][0], acc[mt][2 * q][1], acc[mt][2 * q + 1][0], acc[mt][2 * q + 1][1]};
        }
    }
    if (a.silu) {
        __syncthreads();
        for (int it = tid; it < PR * 16; it += NT) {
            const int row = it >> 4, c4 = it & 15;
            if (m_base + row < a.M) *(f32x4*)(P + (size_t)(m_base + row) * a.ldp + tn * 64 + 4 * c4) = *(const f32x4*)(T + row * TS + 4 * c4);
        }
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
                                                              const float* __restrict__ nw, float eps, signed char* __restrict__ xq, float* __restrict__ xd, int xs) {
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
    if (n4 <= 1024) {
        // rows of <= 4096 outputs (o_proj / down_proj of the 8B widths): a thread's four chunks stay in registers through the three passes
        // -- the same values in the same order as below, without reading the row back twice, and every slice load of the row in flight
        // at once (8.1 -> ~6 us per launch at 8 slices)
        f32x4 o[4], wv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k4 = t2 + 256 * j;
            o[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; wv[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (k4 >= n4) continue;
            if (NORM) wv[j] = *(const f32x4*)(nw + (k4 << 2));
            if (EPI == EPI_SILUMUL) {
                const f32x4 a0 = reduce4(8 * k4), a1 = reduce4(8 * k4 + 4);
                o[j][0] = (a0[0] / (1.0f + expf(-a0[0]))) * a0[1]; o[j][1] = (a0[2] / (1.0f + expf(-a0[2]))) * a0[3];
                o[j][2] = (a1[0] / (1.0f + expf(-a1[0]))) * a1[1]; o[j][3] = (a1[2] / (1.0f + expf(-a1[2]))) * a1[3];
            } else {
                const f32x4 v = reduce4(4 * k4), c = *(const f32x4*)(yr + 4 * k4);
                o[j] = (f32x4){v[0] + c[0], v[1] + c[1], v[2] + c[2], v[3] + c[3]};
            }
            *(f32x4*)(yr + 4 * k4) = o[j];
        }
        float r = 1.f;
        if (NORM) {
            float ss = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (t2 + 256 * j < n4) ss = fmaf(o[j][3], o[j][3], fmaf(o[j][2], o[j][2], fmaf(o[j][1], o[j][1], fmaf(o[j][0], o[j][0], ss))));
            const float t = wave_sum(ss);
            if (lane == 0) red[w2] = t;
            __syncthreads();
            r = 1.0f / sqrtf(((red[0] + red[1]) + (red[2] + red[3])) / (float)Kout + eps);
        }
        uint32_t* xqr = (uint32_t*)(xq + (size_t)m * Kout);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k4 = t2 + 256 * j;
            if (k4 >= n4) break;
            f32x4 xv = o[j];
            if (NORM) {
                xv[0] = __fmul_rn(__fmul_rn(xv[0], r), wv[j][0]); xv[1] = __fmul_rn(__fmul_rn(xv[1], r), wv[j][1]);
                xv[2] = __fmul_rn(__fmul_rn(xv[2], r), wv[j][2]); xv[3] = __fmul_rn(__fmul_rn(xv[3], r), wv[j][3]);
            }
            float am = fmaxf(fmaxf(fabsf(xv[0]), fabsf(xv[1])), fmaxf(fabsf(xv[2]), fabsf(xv[3])));
            am = fmaxf(am, __shfl_xor(am, 1)); am = fmaxf(am, __shfl_xor(am, 2)); am = fmaxf(am, __shfl_xor(am, 4));
            const float d = am / 127.0f;
            const float id = d != 0.f ? 1.0f / d : 0.f;
            uint32_t pk = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk |= ((uint32_t)(int)roundf(xv[e] * id) & 0xFFu) << (8 * e);
            xqr[k4] = pk;
            if ((t2 & 7) == 0) xd[(size_t)(k4 >> 3) * xs + m] = f16r(d);
        }
        return;
    }
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
            if ((t2 & 7) == 0) xd[(size_t)(k4 >> 3) * xs + m] = f16r(d);
        }
    }
}

template <int EPI>
static void launch_q8_epilogue_quant(const float* ws, int M, int N, int ks, float* y, int ldy, const QNext& nx, int xs, hipStream_t s) {
#define CM_QQ(KS_) do { if (nx.nw) hipLaunchKernelGGL((q8_splitk_quant_kernel<EPI, KS_, true>), dim3(M), dim3(256), 0, s, ws, M, N, ks, y, ldy, nx.nw, nx.eps, nx.xq, nx.xd, xs); \
                        else hipLaunchKernelGGL((q8_splitk_quant_kernel<EPI, KS_, false>), dim3(M), dim3(256), 0, s, ws, M, N, ks, y, ldy, nx.nw, nx.eps, nx.xq, nx.xd, xs); } while (0)
    switch (ks) {
        case 1: CM_QQ(1); break; case 2: CM_QQ(2); break; case 3: CM_QQ(3); break; case 4: CM_QQ(4); break; case 5: CM_QQ(5); break;
        case 6: CM_QQ(6); break; case 8: CM_QQ(8); break; case 10: CM_QQ(10); break; case 12: CM_QQ(12); break; case 16: CM_QQ(16); break;
        default: CM_QQ(0);
    }
#undef CM_QQ
}

template <int EPI>
static void launch_q8_epilogue(const float* ws, int M, int N, int ks, float* y, int ldy, hipStream_t s) {
    const int eb = (int)std::min<size_t>(((size_t)M * (N / 4) + 255) / 256, 2048);
#define CM_QE(KS_) hipLaunchKernelGGL((q8_splitk_epilogue_kernel<EPI, KS_>), dim3(eb), dim3(256), 0, s, ws, M, N, ks, y, ldy)
    switch (ks) {
        case 1: CM_QE(1); break; case 2: CM_QE(2); break; case 3: CM_QE(3); break; case 4: CM_QE(4); break; case 5: CM_QE(5); break;
        case 6: CM_QE(6); break; case 8: CM_QE(8); break; case 10: CM_QE(10); break; case 12: CM_QE(12); break; case 16: CM_QE(16); break;
        default: CM_QE(0);
    }
#undef CM_QE
}

bool gemm_q8_ok(const QWeight& w, int M) {
    return (w.fmt == QFMT_Q8_0 || w.fmt == QFMT_Q4_K || w.fmt == QFMT_Q6_K) && M >= 1 && M <= QGEMM_BIGM && w.N % 128 == 0 && w.K % (32 * QG_MIN) == 0;
}

// y (+)= dequant(W) . dequant(xq)^T over the group's rows; epi = EPI_STORE | EPI_RESADD | EPI_SILUMUL (GEMV epilogue codes).
// EPI_STORE with a row stride the workspace cannot hold (the vocabulary head) is written in place by an unsplit launch.
// Host-side plan of one launch (pure: no device calls; cm_debug_qgemm_plan exposes it to the CPU tests).
//   geometry: M <= 32 -> 4 waves x 1 m-tile, M <= 64 -> 4 waves x 2 (groups of 4 blocks, two workgroups per CU); above that 8 waves (two halves of the rows) x 2
//   m-tiles, one workgroup per CU, groups of 4 blocks (CM_QGEMM_GEO = 3, the default; Q8_0 serving of Qwen3-8B at 128 sequences 11.46 K
//   tok/s when it was chosen), or the same with groups of 8 blocks (= 0: 11.01 K), or 4 waves x 4 m-tiles with groups of 4 blocks: 80 KB of
//   LDS, two independent workgroups per CU (= 1: 11.18 K);
//   K split: the chip holds `cap` workgroups at a time (256 registers per lane: 2 waves per SIMD); a launch of `tiles * ks` of them runs in
//   ceil(tiles ks / cap) rounds of (1 start-up + ceil(G / ks) groups), and every slice costs a write + a read of M x N f32.
QGemmPlan plan_gemm_q8(int M, int N, int K, int epi, bool have_ws, size_t ws_floats, int num_cu, int fmt) {
    QGemmPlan p{};
    if (M < 1 || M > QGEMM_BIGM || N % 128 != 0 || K % (32 * QG_MIN) != 0 || (epi != EPI_STORE && epi != EPI_RESADD && epi != EPI_SILUMUL)) return p;
    static const int geo_env = getenv("CM_QGEMM_GEO") ? atoi(getenv("CM_QGEMM_GEO")) : 3;
    // (more than 128 rows -- a prompt pass: geometry 4 = 8 waves x 4 m-tiles, 256 rows per workgroup, ceil(M / 256) m-panels per launch)
    const int geo = M > QGEMM_MAXM ? 4 : M > 64 ? (geo_env == 0 ? 1 : geo_env == 1 ? 2 : 3) : 0;
    p.geo = geo;
    // (<= 64 rows, 4 waves: groups of 4 blocks = 52 / 61 KB of LDS, TWO workgroups per CU as `cap` below assumes; with groups of 8 -- 104 KB --
    // a CU held one 4-wave workgroup, one wave per SIMD.  CM_QGEMM_QG_SMALL = 8: A/B)
    static const int qg_small = getenv("CM_QGEMM_QG_SMALL") && atoi(getenv("CM_QGEMM_QG_SMALL")) == 8 ? 8 : 4;
    p.mh = geo == 1 || geo == 3 || geo == 4 ? 2 : 1; p.mt = geo == 2 || geo == 4 ? 4 : M > 32 ? 2 : 1; p.qg = geo >= 2 ? 4 : geo == 1 ? 8 : qg_small;
    if (fmt == QFMT_Q4_K || fmt == QFMT_Q6_K) {      // groups of half a 256-block (Q4_K: 4 sub-blocks + 1 virtual block; Q6_K: 8 sub-blocks of 16); geometries <1,1> <1,2> <2,2> (Q6_K: also <2,4>)
        p.qg = 4;
        if (geo != 4) { p.mh = M > 64 ? 2 : 1; p.mt = M > 32 ? 2 : 1; }
        else if (fmt == QFMT_Q4_K) { p.mh = 2; p.mt = 2; }       // (its panels carry two weight planes: m-panels of 128 rows)
        // Q4_K: 78 KB of weight planes per workgroup = one workgroup per CU whatever the rows: 8 waves also for <= 64 rows (two halves x one
        // m-tile) -- four waves alone (one per SIMD) cannot hide the panel latency (64-row round of 4 layers: 1013 -> 734 us)
        if (fmt == QFMT_Q4_K && geo != 4 && M <= 64) { p.mh = 2; p.mt = 1; }       // (also below 33 rows, where the second half multiplies padding: 16-row round of 4 layers 816 -> 693 us)
    }
    const int pr = 32 * p.mt * p.mh;
    p.mpan = geo == 4 ? (M + pr - 1) / pr : 1;
    const int nkb_all = K >> 5, tiles = (N / 128) * p.mpan, G = nkb_all / p.qg;
    p.groups = G;
    p.lds = (size_t)(2 * p.qg * std::max(pr, (int)QGEMM_MAXM) + 4 * p.mh * p.qg * 32) * sizeof(float) + (size_t)2 * (128 + pr) * (p.qg * 32 + 16);
    if (fmt == QFMT_Q4_K)
        p.lds = (size_t)(2 * QKB * QGEMM_MAXM + 4 * p.mh * 64) * sizeof(float) + (size_t)2 * 128 * 304 + (size_t)2 * pr * (QKROW + 16);
    if (fmt == QFMT_Q6_K)
        p.lds = (size_t)(2 * std::max(pr, (int)QGEMM_MAXM) + 4 * p.mh * 8 * 32) * sizeof(float) + (size_t)2 * (128 + pr) * 136;
    // (the SiLU tile of an unsplit gate|up launch lives in the panels: [rows][68] floats)
    p.lds = std::max(p.lds, (size_t)(2 * QKB * std::max(pr, (int)QGEMM_MAXM) + 4 * p.mh * QKB * 32) * sizeof(float) + (size_t)pr * 68 * 4);
    const int cap = num_cu * (p.mh == 2 ? 1 : 2);
    const double tgroup_us = geo == 4 ? 4.0 : 2.0, fill_us = 2.5, part_us = 8.0 * M * N / 3.0e6;        // (partials at ~3 TB/s, write + read)
    int ks = 1;
    double best = 1e30;
    const bool direct_only = epi == EPI_STORE && (size_t)M * N > ws_floats;            // (the vocabulary head: written in place, unsplit)
    for (int k = 1; k <= 16 && k <= G && !direct_only; ++k) {
        if (k == 7 || k == 9 || k == 11 || (k > 12 && k < 16)) continue;                  // (splits the reduction kernels are unrolled for)
        if (k > 1 && (!have_ws || (size_t)k * M * N > ws_floats)) break;
        const int rounds = (tiles * k + cap - 1) / cap;
        const double c = rounds * (fill_us + tgroup_us * ((G + k - 1) / k)) + (k > 1 || epi == EPI_RESADD ? k * part_us : 0.0);
        if (c < best) { best = c; ks = k; }
    }
    static const int ks_env = getenv("CM_QGEMM_KS") ? atoi(getenv("CM_QGEMM_KS")) : 0;           // tuning: force the split
    if (ks_env > 0 && !direct_only && ks_env <= G && have_ws && (size_t)ks_env * M * N <= ws_floats) ks = ks_env;
    // unsplit: the store (and SiLU(gate) * up of a gate|up projection) happens in the GEMM's own epilogue, no partial slices
    p.direct = (epi == EPI_STORE || epi == EPI_SILUMUL) && (direct_only || ks == 1);
    // the partial slices of a residual / split projection always go through the workspace: refuse what it cannot hold (the
    // caller falls back to the batched GEMV) instead of writing past it
    if (!p.direct && (!have_ws || (size_t)ks * M * N > ws_floats)) return p;
    p.ks = p.direct ? 1 : ks;
    p.grid = tiles * p.ks;
    p.ok = p.lds <= (size_t)160 * 1024;
    return p;
}

bool launch_gemm_q8(const QGemmArgs& a0, int epi, float* y, int ldy, float* ws, size_t ws_floats, int num_cu, hipStream_t s,
                    const QNext* next, int* fused, QDefer* defer) {
    QGemmArgs a = a0;
    if (defer) { defer->ks = 1; defer->slice = 0; defer->ws = nullptr; }
    if (fused) *fused = 0;
    if (!gemm_q8_ok(a.w, a.M)) return false;
    const bool q4k = a.w.fmt == QFMT_Q4_K, q6k = a.w.fmt == QFMT_Q6_K;
    if (q4k || q6k) next = nullptr;      // (the fused quantisers write Q8_0 blocks; a K-quant consumer takes Q8_K rows: its own quantiser launch)
    const QGemmPlan pl = plan_gemm_q8(a.M, a.w.N, a.w.K, epi, ws != nullptr, ws_floats, num_cu, a.w.fmt);
    if (!pl.ok) return false;
    const int N = a.w.N, tiles = (N / 128) * pl.mpan, mh = pl.mh, mt = pl.mt, geo = pl.geo;
    a.mpan = pl.mpan;
    if (a.xs <= 0) a.xs = QGEMM_MAXM;
    if (a.xs < std::min(a.M, pl.mpan * 32 * mt * mh) && pl.mpan > 1) return false;       // (the scale rows of a multi-panel launch must hold every row read)
    const int xs = a.xs;
    int ks = pl.ks;
    const bool direct = pl.direct;
    const size_t lds = pl.lds;
    a.silu = 0; a.nxq = nullptr; a.nxd = nullptr;
    static const int sq_env = getenv("CM_QGEMM_SILUQ") ? atoi(getenv("CM_QGEMM_SILUQ")) : 1;       // A/B: the unsplit gate|up GEMM quantises its own rows
    if (direct) {
        ks = 1; a.ws = y; a.ldp = ldy; a.slice = 0; a.silu = epi == EPI_SILUMUL;
        if (a.silu && sq_env && next != nullptr && next->nw == nullptr && next->xq2 != nullptr && (N / 2) % 32 == 0) { a.nxq = next->xq2; a.nxd = next->xd2; }
    }
    else { a.ws = ws; a.ldp = N; a.slice = (size_t)a.M * N; }
    a.ksplit = ks;
    static DevOnce attr;
    attr.run([&] {
        (void)hipFuncSetAttribute((const void*)gemm_q8_i8_kernel<1, 1, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)gemm_q8_i8_kernel<1, 1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)gemm_q8_i8_kernel<1, 2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)gemm_q8_i8_kernel<1, 2, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)gemm_q8_i8_kernel<2, 2, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)gemm_q8_i8_kernel<1, 4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)gemm_q8_i8_kernel<2, 2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)gemm_q8_i8_kernel<2, 4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)gemm_q4k_i8_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)gemm_q4k_i8_kernel<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)gemm_q4k_i8_kernel<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)gemm_q4k_i8_kernel<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)gemm_q6k_i8_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)gemm_q6k_i8_kernel<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)gemm_q6k_i8_kernel<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)gemm_q6k_i8_kernel<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    const dim3 grid(tiles * ks), block(256 * mh);
    if (q6k) {
        if (mh == 2 && mt == 4) hipLaunchKernelGGL((gemm_q6k_i8_kernel<2, 4>), grid, block, lds, s, a);
        else if (mh == 2) hipLaunchKernelGGL((gemm_q6k_i8_kernel<2, 2>), grid, block, lds, s, a);
        else if (mt == 2) hipLaunchKernelGGL((gemm_q6k_i8_kernel<1, 2>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((gemm_q6k_i8_kernel<1, 1>), grid, block, lds, s, a);
    }
    else if (q4k) {
        if (mh == 2 && mt == 1) hipLaunchKernelGGL((gemm_q4k_i8_kernel<2, 1>), grid, block, lds, s, a);
        else if (mh == 2) hipLaunchKernelGGL((gemm_q4k_i8_kernel<2, 2>), grid, block, lds, s, a);
        else if (mt == 2) hipLaunchKernelGGL((gemm_q4k_i8_kernel<1, 2>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((gemm_q4k_i8_kernel<1, 1>), grid, block, lds, s, a);
    }
    else if (geo == 4) hipLaunchKernelGGL((gemm_q8_i8_kernel<2, 4, 4>), grid, block, lds, s, a);
    else if (geo == 3) hipLaunchKernelGGL((gemm_q8_i8_kernel<2, 2, 4>), grid, block, lds, s, a);
    else if (geo == 2) hipLaunchKernelGGL((gemm_q8_i8_kernel<1, 4, 4>), grid, block, lds, s, a);
    else if (mh == 2) hipLaunchKernelGGL((gemm_q8_i8_kernel<2, 2, 8>), grid, block, lds, s, a);
    else if (mt == 2 && pl.qg == 4) hipLaunchKernelGGL((gemm_q8_i8_kernel<1, 2, 4>), grid, block, lds, s, a);
    else if (mt == 2) hipLaunchKernelGGL((gemm_q8_i8_kernel<1, 2, 8>), grid, block, lds, s, a);
    else if (pl.qg == 4) hipLaunchKernelGGL((gemm_q8_i8_kernel<1, 1, 4>), grid, block, lds, s, a);
    else hipLaunchKernelGGL((gemm_q8_i8_kernel<1, 1, 8>), grid, block, lds, s, a);
    if (direct) { if (fused && a.nxq) *fused = 2; return true; }
    if (epi == EPI_STORE && defer != nullptr && ks <= 4) { defer->ks = ks; defer->slice = a.slice; defer->ws = ws; return true; }      // (the consumer adds the slices)
    // the next projection's quantiser rides on the reduction launch (CM_QGEMM_QFUSE = 0: its own launch, A/B); the quantiser's
    // lane map needs whole 32-blocks per 8 lanes: output rows of a multiple of 32 columns
    static const int qfuse_env = getenv("CM_QGEMM_QFUSE") ? atoi(getenv("CM_QGEMM_QFUSE")) : 1;
    const int kout = epi == EPI_SILUMUL ? N / 2 : N;
    if (next != nullptr && qfuse_env != 0 && (epi == EPI_RESADD || epi == EPI_SILUMUL) && kout % 32 == 0) {
        if (epi == EPI_RESADD) launch_q8_epilogue_quant<EPI_RESADD>(ws, a.M, N, ks, y, ldy, *next, xs, s);
        else launch_q8_epilogue_quant<EPI_SILUMUL>(ws, a.M, N, ks, y, ldy, *next, xs, s);
        if (fused) *fused = 1;
        return true;
    }
    if (epi == EPI_STORE) launch_q8_epilogue<EPI_STORE>(ws, a.M, N, ks, y, ldy, s);
    else if (epi == EPI_RESADD) launch_q8_epilogue<EPI_RESADD>(ws, a.M, N, ks, y, ldy, s);
    else launch_q8_epilogue<EPI_SILUMUL>(ws, a.M, N, ks, y, ldy, s);
    return true;
}

}  // namespace cm
