// Prompt-pass GEMM for LARGE M on 256-row tiles fed by LDS-DMA:  C[M, N] (+)= A[M, K] . W[N, K]^T, bf16 operands, f32 accumulate,
// the activation operand as one bf16 plane (SPLIT = 1) or as bf16 hi + lo planes (SPLIT = 2, the parity mode: two MFMAs per
// product).  Same epilogues and the same arithmetic per output element as gemm_bf16_kernel (kernels_prefill.hip) up to
// the order of the k-tiles inside one accumulator, which is identical (k ascending) -- results are bit-equal to that kernel's
// un-split launch.
//
// Why a second kernel.  gemm_bf16_kernel stages a k-tile global -> VGPR -> ds_write_b128 -> LDS.  On gfx950 a ds_write_b128 costs
// 13 LDS cycles per wave-instruction (79 B/clk/CU, MI355X_MICROARCH.md, LDS): a 128 x 256 x 32 tile with hi + lo activations writes
// 32 KB = 416 cycles and reads 384 (12 ds_read_b128 x 8 waves x 4) per k-tile against 1024 cycles of MFMA on the four SIMDs --
// 78 % of the LDS pipe at best overlap, and plain bf16 activations exceed it (568 vs 512): that, not the matrix cores, is what
// the 47 % MFMA-busy of the round-2 profile was waiting for.  Here
//   * tiles arrive by global_load_lds_dwordx4 (LDS-DMA): no staging VGPRs, no ds_write pass.  The LDS image is lane-linear per
//     wave-instruction (16 rows x 64 B), so the bank swizzle is applied to the SOURCE address: lane (row, slot p) fetches the
//     16-byte k-chunk p ^ swz(row) of its row, and the fragment reads apply the same involution (cdna_hip_programming.md rule 21);
//   * three LDS stages and a COUNTED vmcnt: while tile t is multiplied, tile t + 1 has landed or is landing and tile t + 2 was
//     just requested.  The DMA is issued from inline asm: the compiler's own wait-count pass would otherwise make every ds_read
//     wait for ALL outstanding LDS-DMA (vmcnt(0)) and serialise the pipeline;
//   * the two wave rows of the workgroup ping-pong (see the schedule comment in the kernel): one multiplies from registers while
//     the other fetches, so the matrix cores do not idle through the fetch;
//   * 256 x 256 x 32 per workgroup of 8 waves (2 x 4), 128 x 64 per wave: 20 ds_read_b128 per 64 MFMAs (hi + lo; 12 per 32 plain),
//     half the LDS reads and half the L2 -> CU bytes per flop of the 128 x 256 tile.
#include <cstdlib>

#include "dev_common.h"
#include "kernels.h"

namespace cm {

namespace {

constexpr int TBK = 32;              // k-tile depth (elements): one 64-byte row segment per operand row
constexpr int TST = 3;               // LDS stages

__device__ __forceinline__ int swz256(int row) { return (4 - ((row >> 2) & 3)) & 3; }      // = gemm_swz (kernels_prefill.hip)

// one LDS-DMA piece: every lane's 16 bytes at `gsrc` land at lds_base + 16 * lane.  M0 carries the LDS base and belongs to the
// compiler, so it is saved and restored around the instruction (cdna_hip_programming.md, inline-asm notes).
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_base) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_base)
                 : "memory");
}

}  // namespace

// BN = 256: 8 waves as 2 (M) x 4 (N), a wave owns 128 x 64.  BN = 192: the same with 48 columns per wave (N = 24576 at M = 1024:
// 4 x 128 = 512 tiles = two full rounds of 256 CUs instead of 384 = one and a half).
template <int SPLIT, int EPI, int BN>
__global__ __launch_bounds__(512) void gemm256_kernel(GemmArgs a) {
    constexpr int BM = 256, NI = 8, NJ = BN / 64, WN = BN / 4;           // per wave: NI x NJ tiles of 16 x 16
    constexpr int PLANE = BM * TBK;                                       // elements of one activation plane of a stage
    constexpr int BROWS = BN;                                             // weight rows of a stage
    constexpr int STAGE = SPLIT * PLANE + BROWS * TBK;                    // elements per stage
    constexpr int APIECES = BM / 16 / 8;                                  // 16-row DMA pieces per wave and plane (2)
    constexpr int BPIECES = (BROWS / 16 + 7) / 8;                         // ... of the weight tile (2; BN = 192: 12 pieces, waves 4-7 repeat one)
    constexpr int NLOAD = SPLIT * APIECES + BPIECES;                      // DMA instructions per wave and k-tile
    extern __shared__ __attribute__((aligned(16))) uint16_t lds256[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = a.N / BN;
    const int nb = tiles_m * tiles_n;
    const int ks = (int)blockIdx.x / nb;
    int bid = (int)blockIdx.x % nb;
    if (nb % 8 == 0) bid = (bid % 8) * (nb / 8) + bid / 8;                // blocks sharing a weight tile: consecutive ids, same XCD
    const int tn = bid / tiles_m, tm = bid % tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int K = a.K;
    const int nk_all = K / TBK, kpb = nk_all / a.ksplit;
    const int kbeg = ks * kpb;                                            // this block's k-tiles [kbeg, kbeg + kpb)

    f32x4 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- DMA source addresses of this lane (k-tile 0), one per piece; rows past M are clamped (their products are dropped) ----
    const int prow = lane >> 2, pslot = lane & 3;
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds256;                    // LDS byte offset of the dynamic segment (low half of the flat address)
    const uint16_t* srcA[SPLIT][APIECES];
    const uint16_t* srcB[BPIECES];
    uint32_t dstA[APIECES], dstB[BPIECES];
#pragma unroll
    for (int i = 0; i < APIECES; ++i) {
        const int piece = wave + 8 * i, row = piece * 16 + prow;
        const int grow = min(m0 + row, a.M - 1);
        const size_t off = (size_t)grow * K + (size_t)kbeg * TBK + ((pslot ^ swz256(row)) << 3);
        srcA[0][i] = a.A_hi + off;
        if (SPLIT == 2) srcA[SPLIT - 1][i] = a.A_lo + off;
        dstA[i] = lds0 + (uint32_t)(piece * 16 * TBK) * 2u;
    }
#pragma unroll
    for (int i = 0; i < BPIECES; ++i) {
        int piece = wave + 8 * i;
        if (piece >= BROWS / 16) piece -= 8;                              // BN = 192: waves 4-7 fetch their first piece twice (equal DMA counts)
        const int row = piece * 16 + prow;
        srcB[i] = a.W + (size_t)(n0 + row) * K + (size_t)kbeg * TBK + ((pslot ^ swz256(row)) << 3);
        dstB[i] = lds0 + (uint32_t)(SPLIT * PLANE + piece * 16 * TBK) * 2u;
    }
    auto issue = [&](int tile, int stage) __attribute__((always_inline)) {          // tile relative to kbeg, clamped by the caller
        const size_t ko = (size_t)tile * TBK;
        const uint32_t so = (uint32_t)(stage * STAGE) * 2u;
#pragma unroll
        for (int i = 0; i < APIECES; ++i) {
            glds16(srcA[0][i] + ko, dstA[i] + so);
            if (SPLIT == 2) glds16(srcA[SPLIT - 1][i] + ko, dstA[i] + so + (uint32_t)PLANE * 2u);
        }
#pragma unroll
        for (int i = 0; i < BPIECES; ++i) glds16(srcB[i] + ko, dstB[i] + so);
    };

    // ---- ping-pong schedule: the two wave rows (waves 0-3 / 4-7; each SIMD hosts one wave of either) run half a k-tile apart.
    // A k-tile is two phases, each closed by the workgroup barrier:
    //   FETCH  : request tile t + 2 by LDS-DMA, read ALL of tile t's fragments into registers, wait for this wave's share of
    //            tile t + 1 (counted vmcnt: tile t + 2 stays in flight), barrier
    //   MULTIPLY: the tile's MFMAs from registers (no memory instruction at all), barrier
    // Wave row 1 enters the loop one barrier late, so while one row multiplies the other fetches: every SIMD's matrix core always
    // has one wave feeding it, and the fetch phase (6 DMA requests + 20 ds_read_b128, ~400 cycles) hides under the partner's 1024
    // MFMA cycles.  In lock step (both rows fetching, then both multiplying: the first version of this kernel) the matrix cores
    // idled through every fetch: 605 TFLOP/s useful on the 1024-row gate||up instead of 552 for the register-staged kernel.
    // Hazards, with three stages: the DMA of tile t + 2 overwrites the stage of tile t - 1, whose last reader (the late row's
    // FETCH of tile t - 1) finished before the barrier the early row passed to get here; a row's FETCH of tile t + 1 comes after a
    // barrier that every wave passed AFTER waiting for its own share of tile t + 1.
    const int fr = lane & 15, fk = (((lane >> 4) ^ swz256(fr)) << 3);
    bf16x8 ah[NI], al[NI], bfrag[NJ];
    auto fetch = [&](int stage) __attribute__((always_inline)) {
        const uint16_t* S = lds256 + stage * STAGE;
        const uint16_t* Bs = S + SPLIT * PLANE;
#pragma unroll
        for (int j = 0; j < NJ; ++j) bfrag[j] = *(const bf16x8*)&Bs[(wc * WN + j * 16 + fr) * TBK + fk];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            ah[i] = *(const bf16x8*)&S[(wr * 128 + i * 16 + fr) * TBK + fk];
            if (SPLIT == 2) al[i] = *(const bf16x8*)&S[PLANE + (wr * 128 + i * 16 + fr) * TBK + fk];
        }
    };
    auto multiply = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bfrag[j], acc[i][j], 0, 0, 0);
            if (SPLIT == 2) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bfrag[j], acc[i][j], 0, 0, 0);
            }
        }
    };
    // barrier closing a FETCH: at most one k-tile of this wave's DMA outstanding (tile t + 1 has landed), fragment reads done
    auto turn_fetch = [&]() __attribute__((always_inline)) {
        if constexpr (NLOAD == 6) asm volatile("s_waitcnt vmcnt(6)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else if constexpr (NLOAD == 4) asm volatile("s_waitcnt vmcnt(4)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else static_assert(NLOAD == 4 || NLOAD == 6, "DMA count per k-tile");
        __builtin_amdgcn_sched_barrier(0);
    };
    auto turn = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    auto step = [&](int t, int stage) __attribute__((always_inline)) {
        issue(min(t + 2, kpb - 1), (stage + 2) % TST);                    // past the end: a clamped tile into a stage nobody reads
        fetch(stage);
        turn_fetch();
        __builtin_amdgcn_s_setprio(1);                                    // the multiplying wave wins the SIMD's issue slot
        multiply();
        __builtin_amdgcn_s_setprio(0);
        turn();
    };

    issue(0, 0);
    issue(min(1, kpb - 1), 1);
    if constexpr (NLOAD == 6) asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
    if (wr == 1) turn();                                                  // the late row: one barrier behind
    for (int t = 0; t < kpb; t += TST) {
        step(t, 0);
        if (t + 1 < kpb) step(t + 1, 1);
        if (t + 2 < kpb) step(t + 2, 2);
    }
    if (wr == 0) turn();                                                  // (every wave executes the same number of barriers)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // the clamped tail requests: nothing of this wave's DMA may land later

    // ---- epilogue (gemm_bf16_kernel's): C layout of mfma 16x16: col = lane & 15 (n), row = (lane >> 4) * 4 + reg (m) ----
    const int mw = m0 + wr * 128, nw = n0 + wc * WN;
    if (EPI == GEPI_PARTIAL) {
        float* P = a.ws + (size_t)ks * a.M * a.N;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = mw + i * 16 + (lane >> 4) * 4 + r;
                    if (m < a.M) P[(size_t)m * a.N + nw + j * 16 + (lane & 15)] = acc[i][j][r];
                }
        return;
    }
    float bv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) bv[j] = a.bias != nullptr ? a.bias[nw + j * 16 + (lane & 15)] : 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        float cold[NJ][4];
        if (EPI == GEPI_RESADD) {                                         // one batch of clamped loads per row band
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = min(mw + i * 16 + (lane >> 4) * 4 + r, a.M - 1);
                    cold[j][r] = a.C[(size_t)m * a.ldc + nw + j * 16 + (lane & 15)];
                }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = nw + j * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mw + i * 16 + (lane >> 4) * 4 + r;
                float v = acc[i][j][r];
                if (a.bias != nullptr) v += bv[j];
                if (EPI == GEPI_STORE) {
                    if (m < a.M) a.C[(size_t)m * a.ldc + n] = v;
                } else if (EPI == GEPI_RESADD) {
                    if (m < a.M) a.C[(size_t)m * a.ldc + n] = cold[j][r] + v;
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
                } else {                                                  // GEPI_SILUMUL: even column = gate_j, odd column = up_j
                    const float up = dpp_mov<0xB1>(v);                    // lane ^ 1
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

// dynamic LDS of an instantiation (bytes)
static size_t gemm256_lds(int split, int bn) { return (size_t)TST * ((size_t)split * 256 * TBK + (size_t)bn * TBK) * 2; }

template <int SPLIT, int EPI, int BN>
static void launch_one(const GemmArgs& a, int blocks, hipStream_t s) {
    static bool attr = false;
    const size_t lds = gemm256_lds(SPLIT, BN);
    if (!attr) { (void)hipFuncSetAttribute((const void*)gemm256_kernel<SPLIT, EPI, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
    hipLaunchKernelGGL((gemm256_kernel<SPLIT, EPI, BN>), dim3(blocks), dim3(512), lds, s, a);
}

// a.ksplit set by the caller (1: epilogue `epi`; > 1: GEPI_PARTIAL tiles, the caller runs gemm_splitk_epilogue_kernel)
bool launch_gemm256(const GemmArgs& a, int epi, int bn, hipStream_t s) {
    if ((bn != 256 && bn != 192) || a.N % bn != 0 || a.K % TBK != 0 || (a.K / TBK) % a.ksplit != 0) return false;
    const int blocks = ((a.M + 255) / 256) * (a.N / bn) * a.ksplit;
    const bool split = a.A_lo != nullptr;
    const int e = a.ksplit > 1 ? (int)GEPI_PARTIAL : epi;
#define CM_G256(SP, EP) do { if (bn == 256) launch_one<SP, EP, 256>(a, blocks, s); else launch_one<SP, EP, 192>(a, blocks, s); } while (0)
#define CM_G256_EPI(SP) do { if (e == GEPI_STORE) CM_G256(SP, GEPI_STORE); else if (e == GEPI_RESADD) CM_G256(SP, GEPI_RESADD); \
        else if (e == GEPI_ACT_SPLIT) CM_G256(SP, GEPI_ACT_SPLIT); else if (e == GEPI_SILUMUL) CM_G256(SP, GEPI_SILUMUL); \
        else CM_G256(SP, GEPI_PARTIAL); } while (0)
    if (split) CM_G256_EPI(2); else CM_G256_EPI(1);
#undef CM_G256_EPI
#undef CM_G256
    return true;
}

}  // namespace cm
