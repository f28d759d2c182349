// Prompt-pass GEMM, round 6: ONE wave per SIMD, 128 x 128 of accumulators per wave (the design the round-4 ablations of
// kernels_gemm256.hip pointed at, DESIGN 3.6: its two-row ping-pong tops out at ~72 % of the matrix-core peak with no memory traffic at
// all and each LDS path costs ~15 % on top).  C[M, N] (+)= A[M, K] . W[N, K]^T, bf16 operands, f32 accumulate, activations as one bf16
// plane (SPLIT = 1) or bf16 hi + lo planes (SPLIT = 2, the parity mode: two MFMAs per product).
//
//   * 4 waves as 2 (M) x 2 (N) on a 256 x BN tile (plain) / 128 x BN x two planes (parity): a wave owns 128 x 128 outputs (parity:
//     64 rows x two planes x 128) = 16 tiles of v_mfma_f32_32x32x16_bf16 per 16-deep k-step -- 8 fragment reads (4 A + 4 B, 1 KB each)
//     per 16 MFMAs, against 12 reads per 16 MFMA-equivalents of the 8-wave 128 x 64 layout: a third less LDS traffic per flop, and no
//     second wave on the SIMD to take turns with: the software pipeline lives inside the wave;
//   * tiles arrive by LDS-DMA in whole 128-byte rows exactly like kernels_gemm256.hip (global_load_lds_dwordx4, 8 rows per
//     wave-instruction, bank swizzle on the SOURCE address: chunk ^ ((row >> 1) & 7) -- conflict-free for the 32-row fragments of the
//     32 x 32 instruction too: the 16 lanes of every ds_read_b128 group fall on 16 different 16-byte slots), two LDS stages of 64 KB;
//   * schedule per 64-deep k-tile (4 k-steps of 16 MFMAs): the fragments of step s + 1 are read while step s multiplies (two register
//     sets); ONE workgroup barrier per k-tile, placed between the fragment reads of the last step and its MFMAs: behind it every wave
//     holds the tile's last fragments in registers, so the tile's stage is free -- the DMA of tile t + 2 is issued into it, one piece
//     per MFMA of that last step, with a whole k-tile (> 2000 cycles) to land -- and tile t + 1 (waited for with vmcnt(0) just before
//     the barrier) is readable: its first fragments are fetched under the same 16 MFMAs.  The matrix core never waits for LDS or a
//     barrier with an empty queue except for the skew of the four waves at that one barrier.
// Epilogues and the fixed-order split-K are those of gemm_bf16_kernel / gemm256_kernel; accumulation order per output: k ascending,
// hi then lo inside a k-step (the 32 x 32 x 16 instruction sums 16 products per step where the 16 x 16 x 32 one sums 32: logits agree
// with the other kernels to f32 summation order, tests 2e-5).
#include <algorithm>
#include <cstdlib>

#include "dev_common.h"
#include "kernels.h"

namespace cm {

namespace {

constexpr int WBK = 64;              // k-tile depth (elements): one 128-byte line per operand row
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8v;

__device__ __forceinline__ void glds16w(const void* gsrc, uint32_t lds_base) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_base)
                 : "memory");
}

__device__ __forceinline__ f32x16 mma32(const bf16x8& a, const bf16x8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8v, a), __builtin_bit_cast(bf16x8v, b), c, 0, 0, 0);
}

}  // namespace

template <int SPLIT, int EPI, int BN>
__global__ __launch_bounds__(256) void gemmw4_kernel(GemmArgs a) {
    constexpr int BM = SPLIT == 2 ? 128 : 256;
    constexpr int WM = BM / 2, WN = BN / 2;                    // a wave's outputs
    constexpr int NI = WM / 32, NJ = WN / 32;                  // 32 x 32 tiles per wave and plane
    constexpr int NA = SPLIT * NI;                             // A fragments per k-step (hi tiles, then lo tiles)
    constexpr int PLANE_B = BM * WBK * 2, ASTAGE_B = SPLIT * PLANE_B, WSTAGE_B = BN * WBK * 2, STAGE_B = ASTAGE_B + WSTAGE_B;   // bytes
    constexpr int NPC = (SPLIT * BM + BN) / 8;                 // 8-row DMA pieces (1 KB) of a stage
    constexpr int NLOAD = NPC / 4;                             // per wave
    static_assert(NPC % 4 == 0 && NLOAD <= 16 && NLOAD % 2 == 0 && (BM / 8) % 4 == 0, "piece distribution");
    constexpr int NMF = NA * NJ;                               // MFMAs of a k-step (16; 12 at BN = 192)
    extern __shared__ __attribute__((aligned(16))) uint16_t ldsw4[];
    const char* const ldsc = (const char*)ldsw4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = a.N / BN;
    const int nb = tiles_m * tiles_n;
    const int ks = (int)blockIdx.x / nb;
    int bid = (int)blockIdx.x % nb;
    if (nb % 8 == 0) bid = (bid % 8) * (nb / 8) + bid / 8;     // blocks sharing a weight tile: consecutive ids, same XCD
    const int tn = bid / tiles_m, tm = bid % tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int K = a.K;
    const int nk_all = K / WBK, kpb = nk_all / a.ksplit;
    const int kbeg = ks * kpb;

    f32x16 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- DMA pieces of this wave: piece p = wave + 4 g covers 8 rows x 128 B of plane hi, plane lo or the weight tile.  The loop is
    // ISSUE-bound (one wave per SIMD: ~7 slots per 32-cycle MFMA; every scalar or vector instruction next to the MFMAs was measurable:
    // +5 % for two debug branches per step), so a request is three instructions: the LDS base into M0, the hazard nop, and
    // global_load_lds with a SCALAR 64-bit base (operand plane + k-tile: advanced once per tile) and a 32-bit lane offset that never
    // changes (row, clamped to M - 1, and the source-side swizzle) ----
    const int prow = lane >> 3, pslot = lane & 7;
    const uint32_t lds0 = (uint32_t)(uintptr_t)ldsw4;
    uint32_t voff[NLOAD];                                      // byte offset of this lane's 16 bytes from the operand's base
    uint32_t dst[NLOAD];                                       // LDS byte address of the piece in stage 0
#pragma unroll
    for (int g = 0; g < NLOAD; ++g) {
        const int p = wave + 4 * g;
        const int plane = p < SPLIT * (BM / 8) ? p / (BM / 8) : SPLIT;        // SPLIT: the weight tile
        const int r0 = (plane < SPLIT ? p - plane * (BM / 8) : p - SPLIT * (BM / 8)) * 8, row = r0 + prow;
        const uint32_t kc = (uint32_t)((pslot ^ ((row >> 1) & 7)) << 4);
        voff[g] = (uint32_t)((plane < SPLIT ? (size_t)min(m0 + row, a.M - 1) : (size_t)(n0 + row)) * K * 2) + kc;
        dst[g] = lds0 + (uint32_t)(plane < SPLIT ? plane * PLANE_B + r0 * WBK * 2 : ASTAGE_B + r0 * WBK * 2);
    }
    const char* const gA_hi = (const char*)a.A_hi + (size_t)kbeg * WBK * 2;
    const char* const gA_lo = SPLIT == 2 ? (const char*)a.A_lo + (size_t)kbeg * WBK * 2 : gA_hi;
    const char* const gW = (const char*)a.W + (size_t)kbeg * WBK * 2;
    // two pieces per statement (M0 saved and restored once)
    auto pieces2 = [&](int g, int t) __attribute__((always_inline)) {
        const int tt = min(t, kpb - 1);                        // past the end: the last tile again, into a stage nobody reads
        const size_t ko = (size_t)tt * (WBK * 2);
        const uint32_t so = (uint32_t)((t & 1) * STAGE_B);
        auto base_of = [&](int gg) __attribute__((always_inline)) {
            // (piece wave + 4 gg: BM / 8 is a multiple of 4, so the operand depends on gg alone -- a compile-time choice)
            return (4 * gg < BM / 8 ? gA_hi : 4 * gg < SPLIT * (BM / 8) ? gA_lo : gW) + ko;
        };
        const char* b0 = base_of(g);
        const char* b1 = base_of(g + 1);
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\t"
                     "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %6\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff[g]), "v"(voff[g + 1]), "s"(dst[g] + so), "s"(dst[g + 1] + so), "s"(b0), "s"(b1)
                     : "memory");
    };

    // ---- fragments: lane (fr, fh) reads the 16 bytes k = 16 s + 8 fh ... of row fr of a 32-row tile; every read is base + immediate:
    // the lane part (row, swizzled chunk of step s) and the stage are folded into four A and four B base registers per k-tile ----
    const int fr = lane & 31, fh = lane >> 5, sw = (fr >> 1) & 7;
    uint32_t swz[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) swz[s] = (uint32_t)(((2 * s + fh) ^ sw) << 4);
    const uint32_t arow = (uint32_t)((wr * WM + fr) * WBK * 2), brow = (uint32_t)(ASTAGE_B + (wc * WN + fr) * WBK * 2);
    bf16x8 fa[2][NA], fb[2][NJ];
    // all fragments of k-step `s` whose lane bases are ab / bb, number f of NA + NJ, into register set `set`
    auto frag = [&](uint32_t ab, uint32_t bb, int f, int set) __attribute__((always_inline)) {
        if (f < NA) fa[set][f] = *(const bf16x8*)(ldsc + ab + (f >= NI ? PLANE_B : 0) + (f % NI) * (32 * WBK * 2));
        else fb[set][f - NA] = *(const bf16x8*)(ldsc + bb + (f - NA) * (32 * WBK * 2));
    };
    // MFMA number q of a k-step: hi tiles first (i major, j minor), then the lo tiles -- an accumulator is revisited NI * NJ MFMAs later
    auto mfma_q = [&](int q, int set) __attribute__((always_inline)) {
        const int ia = q / NJ, j = q % NJ, i = ia % NI;
        acc[i][j] = mma32(fa[set][ia], fb[set][j], acc[i][j]);
    };

    // prologue: tiles 0 and 1 requested, tile 0 waited for, its first fragments read
#pragma unroll
    for (int g = 0; g < NLOAD; g += 2) pieces2(g, 0);
#pragma unroll
    for (int g = 0; g < NLOAD; g += 2) pieces2(g, 1);
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NLOAD) : "memory");
#pragma unroll
    for (int f = 0; f < NA + NJ; ++f) frag(arow + swz[0], brow + swz[0], f, 0);

    for (int t = 0; t < kpb; ++t) {
        const uint32_t sb = (uint32_t)((t & 1) * STAGE_B), sn = (uint32_t)(STAGE_B - sb);
        const uint32_t a_cur = arow + sb, b_cur = brow + sb;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int set = s & 1, nset = set ^ 1;
            if (s < 3) {
                // the next step's fragments (same tile) under this step's MFMAs: one read in front of every second MFMA
                const uint32_t ab = a_cur + swz[s + 1], bb = b_cur + swz[s + 1];
#pragma unroll
                for (int q = 0; q < NMF; ++q) {
                    if ((q & 1) == 0 && (q >> 1) < NA + NJ) frag(ab, bb, q >> 1, nset);
                    mfma_q(q, set);
                    if (q & 1) __builtin_amdgcn_sched_barrier(0);
                }
                // (fragments that did not fit the pairs: 8 reads need 16 MFMAs; 7 reads / 12 MFMAs at BN = 192 do not)
#pragma unroll
                for (int f = NMF / 2; f < NA + NJ; ++f) frag(ab, bb, f, nset);
            } else {
                // last step of the tile: every wave holds its fragments in registers -> the tile's stage is free, tile t + 1 has landed
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                // (after the last tile these reads fetch a stale stage nobody multiplies: cheaper than a branch per read)
                const uint32_t ab = arow + sn + swz[0], bb = brow + sn + swz[0];
#pragma unroll
                for (int q = 0; q < NMF; ++q) {
                    if ((q & 1) == 0 && (q >> 1) < NA + NJ) frag(ab, bb, q >> 1, nset);
                    mfma_q(q, set);
                    __builtin_amdgcn_sched_barrier(0);
                    if ((q & 1) && q - 1 < NLOAD) { pieces2(q - 1, t + 2); __builtin_amdgcn_sched_barrier(0); }      // two requests behind every second MFMA
                }
#pragma unroll
                for (int f = NMF / 2; f < NA + NJ; ++f) frag(ab, bb, f, nset);
#pragma unroll
                for (int g = NMF; g < NLOAD; g += 2) pieces2(g, t + 2);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the clamped tail requests: no DMA of this wave may land after it exits

    // ---- epilogue.  C layout of the 32 x 32 instruction: column n = lane & 31, rows m = 8 (r / 4) + 4 (lane >> 5) + r % 4 ----
    const int mw = m0 + wr * WM, nw = n0 + wc * WN;
    const int ln = lane & 31, lm = 4 * (lane >> 5);
    if (EPI == GEPI_PARTIAL) {
        float* P = a.ws + (size_t)ks * a.M * a.N;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mw + i * 32 + 8 * (r >> 2) + lm + (r & 3);
                    if (m < a.M) P[(size_t)m * a.N + nw + j * 32 + ln] = acc[i][j][r];
                }
        return;
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = nw + j * 32 + ln;
            const float bv = a.bias != nullptr ? a.bias[n] : 0.f;
            float cold[16];
            if (EPI == GEPI_RESADD) {                          // one batch of clamped loads per tile
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = min(mw + i * 32 + 8 * (r >> 2) + lm + (r & 3), a.M - 1);
                    cold[r] = a.C[(size_t)m * a.ldc + n];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mw + i * 32 + 8 * (r >> 2) + lm + (r & 3);
                float v = acc[i][j][r];
                if (a.bias != nullptr) v += bv;
                if (EPI == GEPI_STORE) {
                    if (m < a.M) a.C[(size_t)m * a.ldc + n] = v;
                } else if (EPI == GEPI_RESADD) {
                    if (m < a.M) a.C[(size_t)m * a.ldc + n] = cold[r] + v;
                } else if (EPI == GEPI_ACT_SPLIT) {
                    if (m < a.M) {
                        float h = v;
                        if (a.act == 1) h = 0.5f * v * (1.0f + tanhf(0.7978845608028654f * (v + 0.044715f * v * v * v)));
                        else if (a.act == 2) h = 0.5f * v * (1.0f + erff(v * 0.7071067811865476f));
                        const size_t off = (size_t)m * a.N + n;
                        const uint16_t hh = f32_to_bf16(h);
                        a.H_hi[off] = hh;
                        if (a.H_lo) a.H_lo[off] = f32_to_bf16(h - bf16_to_f32(hh));
                    }
                } else {                                       // GEPI_SILUMUL: even column = gate_j, odd column = up_j
                    const float up = dpp_mov<0xB1>(v);         // lane ^ 1
                    if (((lane & 1) == 0) && m < a.M) {
                        const float h = (v / (1.0f + expf(-v))) * up;
                        const size_t off = (size_t)m * (a.N / 2) + (n >> 1);
                        const uint16_t hh = f32_to_bf16(h);
                        a.H_hi[off] = hh;
                        if (a.H_lo) a.H_lo[off] = f32_to_bf16(h - bf16_to_f32(hh));
                    }
                }
            }
        }
    }
}

static size_t gemmw4_lds(int split, int bn) { return (size_t)2 * ((size_t)split * (split == 2 ? 128 : 256) * WBK + (size_t)bn * WBK) * 2; }

template <int SPLIT, int EPI, int BN>
static void launch_w4(const GemmArgs& a, int blocks, hipStream_t s) {
    static DevOnce attr;
    const size_t lds = gemmw4_lds(SPLIT, BN);
    attr.run([&] { (void)hipFuncSetAttribute((const void*)gemmw4_kernel<SPLIT, EPI, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
    hipLaunchKernelGGL((gemmw4_kernel<SPLIT, EPI, BN>), dim3(blocks), dim3(256), lds, s, a);
}

int gemmw4_rows(bool split) { return split ? 128 : 256; }

// a.ksplit set by the caller (1: epilogue `epi`; > 1: GEPI_PARTIAL tiles, the caller runs gemm_splitk_epilogue_kernel)
bool launch_gemmw4(const GemmArgs& a, int epi, int bn, hipStream_t s) {
    if ((bn != 256 && bn != 192) || a.N % bn != 0 || a.K % WBK != 0 || (a.K / WBK) % a.ksplit != 0 || a.K / WBK / a.ksplit < 2) return false;
    const bool split = a.A_lo != nullptr;
    const int bm = gemmw4_rows(split);
    const int blocks = ((a.M + bm - 1) / bm) * (a.N / bn) * a.ksplit;
    const int e = a.ksplit > 1 ? (int)GEPI_PARTIAL : epi;
#define CM_W4(SP, EP) do { if (bn == 256) launch_w4<SP, EP, 256>(a, blocks, s); else launch_w4<SP, EP, 192>(a, blocks, s); } while (0)
#define CM_W4_EPI(SP) do { if (e == GEPI_STORE) CM_W4(SP, GEPI_STORE); else if (e == GEPI_RESADD) CM_W4(SP, GEPI_RESADD); \
        else if (e == GEPI_ACT_SPLIT) CM_W4(SP, GEPI_ACT_SPLIT); else if (e == GEPI_SILUMUL) CM_W4(SP, GEPI_SILUMUL); \
        else CM_W4(SP, GEPI_PARTIAL); } while (0)
    if (split) CM_W4_EPI(2); else CM_W4_EPI(1);
#undef CM_W4_EPI
#undef CM_W4
    return true;
}

}  // namespace cm
