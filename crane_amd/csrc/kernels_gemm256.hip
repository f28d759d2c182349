// Prompt-pass GEMM for LARGE M fed by LDS-DMA in FULL cache lines:  C[M, N] (+)= A[M, K] . W[N, K]^T, bf16 operands, f32
// accumulate, the activation operand as one bf16 plane (SPLIT = 1) or as bf16 hi + lo planes (SPLIT = 2, the parity mode: two
// MFMAs per product).  Same epilogues and, per accumulator, the same sequence of MFMAs (k ascending; hi then lo) as
// gemm_bf16_kernel (kernels_prefill.hip): an un-split launch is bit-equal to that kernel's.
//
// Why a second kernel -- what the round-2 profile's 47 % MFMA-busy was waiting for.  gemm_bf16_kernel walks K in 32-element
// k-tiles: every operand row contributes a 64-byte segment per tile, i.e. HALF of a 128-byte cache line, and the other half is
// fetched again from the L2 one k-tile later (the 32 KB vector L1 does not keep a 40 KB tile stream).  The L2 -> L1 fill path
// moves 64 B/clk per CU: a 128 x 256 x 32 tile with hi + lo activations needs 512 rows x 128 B = 64 KB of line fills per
// 1024 MFMA cycles -- 100 % of it.  On top, the tile went global -> VGPR -> ds_write_b128 (13 LDS cycles per wave-instruction,
// MI355X_MICROARCH.md) -> LDS.  A first version of this file kept the 32-deep tiles and only replaced the staging by LDS-DMA and
// a ping-pong of the two wave rows: 552 -> 632 TFLOP/s useful on the 1024-row gate||up, still fill-bound.  This version
//   * walks K in 64-element tiles: an operand row is one whole 128-byte line per tile, fetched once;
//   * brings tiles in by global_load_lds_dwordx4 (LDS-DMA, 8 rows x 128 B per wave-instruction): no staging VGPRs, no ds_write
//     pass.  The LDS image of a DMA is lane-linear, so the bank swizzle sits on the SOURCE address: lane (row, slot p) fetches
//     the 16-byte chunk p ^ ((row >> 1) & 7) of its row and the fragment reads apply the same involution
//     (cdna_hip_programming.md rule 21) -- the 16 lanes of every ds_read_b128 group then fall on 16 different 16-byte slots of
//     the 256-byte bank row;
//   * two LDS stages + the register file as the third, and the two wave rows of the workgroup half a tile apart: one row
//     multiplies from registers while the other fetches the next tile's fragments (schedule comment in the kernel).  The DMA
//     is issued from inline asm: the compiler's own wait-count pass would make every ds_read wait for ALL outstanding LDS-DMA;
//   * parity mode: 128 x BN x 64 tiles (two activation planes: 64 KB per stage at BN = 256); plain bf16: 256 x BN x 64;
//     8 waves as 2 (M) x 4 (N); BN = 256 or 192 (whole rounds of 256 CUs: N = 24576 at M = 1024).
#include <cstdlib>

#include "dev_common.h"
#include "kernels.h"

namespace cm {

namespace {

constexpr int TBK = 64;              // k-tile depth (elements): one 128-byte line per operand row
constexpr int TST = 2;               // LDS stages

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

// WST = 3 (round 4, decode groups: one m-tile, the weights come from HBM, not from a neighbour's L2 lines): a THIRD weight stage, the
// weight requests running two k-tiles ahead (the activation stages stay at two).  With two stages a workgroup has one 32 KB weight
// tile in flight; 96-192 workgroups of a 128-row launch are then 3-6 MB in flight chip-wide -- about 3 TB/s at the ~2 us of a loaded
// HBM round trip, which is what the launch measured.  The waits become counted: the newest weight requests may stay outstanding.
// BMX = 64 (round 4, parity mode): 64-row tiles for decode groups of <= 64 sequences -- half the activation bytes and half the MFMAs
// of a 128-row tile whose upper half would be padding.
template <int SPLIT, int EPI, int BN, int WST = 2, int BMX = 0>
__global__ __launch_bounds__(512) void gemm256_kernel(GemmArgs a) {
    constexpr int BM = BMX ? BMX : (SPLIT == 2 ? 128 : 256);              // rows of a tile (two planes double the activation bytes)
    constexpr int NI = BM / 32, NJ = BN / 64, WM = BM / 2, WN = BN / 4;  // per wave: NI x NJ tiles of 16 x 16
    constexpr int PLANE = BM * TBK;                                       // elements of one activation plane of a stage
    constexpr int ASTAGE = SPLIT * PLANE, WSTAGE = BN * TBK;             // elements of an activation / a weight stage
    constexpr int WBASE = TST * ASTAGE;                                   // LDS image: [TST activation stages][WST weight stages]
    constexpr int APC = BM / 8, NPC = SPLIT * APC + BN / 8;               // 8-row DMA pieces per plane / per stage
    constexpr int NLOAD = (NPC + 7) / 8;                                  // DMA instructions per wave and k-tile (7 or 8)
    static_assert(NPC > 8 * (NLOAD - 1) && NLOAD <= 8, "piece distribution");
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

    // ---- DMA pieces of this wave: piece p = wave + 8 i covers 8 rows x 128 B of plane hi, plane lo or the weight tile.  A wave
    // whose last slot has no piece (NPC not a multiple of 8) repeats its first one: equal DMA counts keep the waits uniform.
    // Rows past M are clamped (their products are dropped in the epilogue). ----
    const int prow = lane >> 3, pslot = lane & 7;
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds256;                    // LDS byte offset of the dynamic segment (low half of the flat address)
    const uint16_t* src[NLOAD];
    uint32_t dst[NLOAD];
    bool isw[NLOAD];
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
        int p = wave + 8 * i;
        if (p >= NPC) p = wave;
        const int plane = p < SPLIT * APC ? p / APC : SPLIT;              // SPLIT: the weight tile
        const int r0 = (plane < SPLIT ? p - plane * APC : p - SPLIT * APC) * 8, row = r0 + prow;
        const size_t kc = (size_t)kbeg * TBK + (size_t)((pslot ^ ((row >> 1) & 7)) << 3);
        if (plane < SPLIT) {
            const uint16_t* base = (SPLIT == 2 && plane == 1) ? a.A_lo : a.A_hi;
            src[i] = base + (size_t)min(m0 + row, a.M - 1) * K + kc;
        } else {
            src[i] = a.W + (size_t)(n0 + row) * K + kc;
        }
        dst[i] = lds0 + (uint32_t)(plane < SPLIT ? plane * PLANE + r0 * TBK : WBASE + r0 * TBK) * 2u;
        isw[i] = plane == SPLIT;
    }
    // weight pieces of this wave (the LAST requests of every issue: the counted waits let exactly these stay in flight)
    constexpr int NWP = (BN / 8) / 8;
    static_assert(WST == 2 || ((SPLIT * APC) % 8 == 0 && (BN / 8) % 8 == 0), "counted waits need whole piece rows per wave");

    const int fr = lane & 15, sw = (fr >> 1) & 7;
    const int fk0 = ((0 + (lane >> 4)) ^ sw) << 3, fk1 = ((4 + (lane >> 4)) ^ sw) << 3;      // element offsets of the lane's chunk, k-steps 0 / 1
    bf16x8 ah[2][NI], al[2][NI], bfr[2][NJ];
    auto fetch = [&](int t) __attribute__((always_inline)) {
        const uint16_t* S = lds256 + (t & 1) * ASTAGE;
        const uint16_t* Bs = lds256 + WBASE + (t % WST) * WSTAGE;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int fk = s ? fk1 : fk0;
#pragma unroll
            for (int j = 0; j < NJ; ++j) bfr[s][j] = *(const bf16x8*)&Bs[(wc * WN + j * 16 + fr) * TBK + fk];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                ah[s][i] = *(const bf16x8*)&S[(wr * WM + i * 16 + fr) * TBK + fk];
                if (SPLIT == 2) al[s][i] = *(const bf16x8*)&S[PLANE + (wr * WM + i * 16 + fr) * TBK + fk];
            }
        }
    };
    // ---- schedule: the two wave rows (waves 0-3 / 4-7; every SIMD hosts one wave of either) run half a tile apart.  A tile is two
    // phases per row, each closed by the workgroup barrier: FETCH (all fragments of tile t: LDS -> registers) and MULTIPLY (the
    // tile's MFMAs from registers).  Row 1 enters one barrier late, so while one row multiplies the other fetches and every
    // SIMD's matrix core always has a wave feeding it.  With the fragments in registers a stage is free once BOTH rows fetched it
    // -- two stages suffice if tile t + 1's DMA is requested by both rows in the SAME slot (the one in which row 0 fetches tile t
    // and row 1 multiplies tile t - 1: the stage of tile t - 1 was fetched by row 1 one slot earlier) and waited for one slot
    // later: row 0 requests its share at the start of its FETCH and waits at the end of its MULTIPLY, row 1 requests its share
    // BETWEEN THE MFMA BANDS of its MULTIPLY (a request costs ~60 cycles of issue there, ~150 in a fetch phase next to the
    // ds_reads: MI355X_MICROARCH.md) and waits at the end of its next FETCH.  Every DMA has a whole slot (> 1000 cycles) to land.
    auto multiply = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < NI; ++i) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[s][i], bfr[s][j], acc[i][j], 0, 0, 0);
                if (SPLIT == 2) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[s][i], bfr[s][j], acc[i][j], 0, 0, 0);
                }
            }
    };
    // requests of one slot: the activation pieces of tile `ta` and the weight pieces of tile `tw` (= ta with two weight stages, ta + 1
    // with three), each clamped to the last tile (past the end: into a stage nobody reads)
    auto piece = [&](int g, int ta, int tw) __attribute__((always_inline)) {
        const int tl = kpb - 1;
        if (isw[g]) { const int tt = min(tw, tl); glds16(src[g] + (size_t)tt * TBK, dst[g] + (uint32_t)((tw % WST) * WSTAGE) * 2u); }
        else { const int tt = min(ta, tl); glds16(src[g] + (size_t)tt * TBK, dst[g] + (uint32_t)((ta & 1) * ASTAGE) * 2u); }
    };
    auto issue_all = [&](int ta, int tw) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < NLOAD; ++g) piece(g, ta, tw);
    };
    // row 1: the MFMAs in 2 * NI row bands, one DMA request of the tile after next behind each of the first NLOAD bands (pinned:
    // left alone the scheduler hoists every MFMA above the requests)
    auto multiply_dma = [&](int ta, int tw) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < NI; ++i) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[s][i], bfr[s][j], acc[i][j], 0, 0, 0);
                if (SPLIT == 2) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[s][i], bfr[s][j], acc[i][j], 0, 0, 0);
                }
                constexpr int PER = (NLOAD + 2 * NI - 1) / (2 * NI);         // requests per band (1 with 128-row tiles; 2 with 64-row tiles)
                const int g = (s * NI + i) * PER;
                if (g < NLOAD) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int p = 0; p < PER; ++p)
                        if (g + p < NLOAD) piece(g + p, ta, tw);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
    };

    // WA: the weight tile requested next to activation tile ta is ta + (WST - 2): one further ahead with the third stage
    constexpr int WA = WST - 2;
    issue_all(0, 0);                                                      // tile 0: every wave its share
    if constexpr (WST == 3) {                                             // ... and the weights of tile 1 (its activations follow in the loop)
#pragma unroll
        for (int g = 0; g < NLOAD; ++g) if (isw[g]) piece(g, 0, 1);
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NWP) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (wr == 0) {
        for (int t = 0; t < kpb; ++t) {
            issue_all(t + 1, t + 1 + WA);
            fetch(t);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            multiply();
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            // this wave's share of tile t + 1 has landed (three stages: its newest weight requests, tile t + 2, may still fly)
            if constexpr (WST == 3) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NWP) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        }
        asm volatile("s_barrier" ::: "memory");                           // (every wave executes the same number of barriers)
    } else {
        issue_all(1, 1 + WA);                                             // its share of tile 1, in the slot in which row 0 fetches tile 0
        asm volatile("s_barrier" ::: "memory");                           // the late row: one barrier behind
        for (int t = 0; t < kpb; ++t) {
            fetch(t);
            if constexpr (WST == 3) asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(NWP) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // its share of tile t + 1 has landed
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            multiply_dma(t + 2, t + 2 + WA);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_barrier" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // the clamped tail requests: no DMA of this wave may land after it exits

    // ---- epilogue (gemm_bf16_kernel's): C layout of mfma 16x16: col = lane & 15 (n), row = (lane >> 4) * 4 + reg (m) ----
    const int mw = m0 + wr * WM, nw = n0 + wc * WN;
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

// rows of a tile / dynamic LDS of an instantiation
// rows of a tile: 64 for parity-mode groups of <= 64 rows (CM_GEMM256_BM64 = 0: never, A/B)
int gemm256_rows(bool split, int M) {
    static const int bm64 = getenv("CM_GEMM256_BM64") ? atoi(getenv("CM_GEMM256_BM64")) : 1;
    return split ? ((M <= 64 && bm64) ? 64 : 128) : 256;
}
static size_t gemm256_lds(int split, int bm, int bn, int wst = 2) { return ((size_t)TST * split * bm * TBK + (size_t)wst * bn * TBK) * 2; }

template <int SPLIT, int EPI, int BN, int WST = 2, int BMX = 0>
static void launch_one(const GemmArgs& a, int blocks, hipStream_t s) {
    static DevOnce attr;
    const size_t lds = gemm256_lds(SPLIT, BMX ? BMX : (SPLIT == 2 ? 128 : 256), BN, WST);
    attr.run([&] { (void)hipFuncSetAttribute((const void*)gemm256_kernel<SPLIT, EPI, BN, WST, BMX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
    hipLaunchKernelGGL((gemm256_kernel<SPLIT, EPI, BN, WST, BMX>), dim3(blocks), dim3(512), lds, s, a);
}

// a.ksplit set by the caller (1: epilogue `epi`; > 1: GEPI_PARTIAL tiles, the caller runs gemm_splitk_epilogue_kernel)
bool launch_gemm256(const GemmArgs& a, int epi, int bn, hipStream_t s) {
    if ((bn != 256 && bn != 192) || a.N % bn != 0 || a.K % TBK != 0 || (a.K / TBK) % a.ksplit != 0) return false;
    const bool split = a.A_lo != nullptr;
    const int bm = gemm256_rows(split, a.M);
    const int blocks = ((a.M + bm - 1) / bm) * (a.N / bn) * a.ksplit;
    const int e = a.ksplit > 1 ? (int)GEPI_PARTIAL : epi;
    // three weight stages (WST = 3): parity-mode tiles (160 KB of LDS at 256 columns, 136 KB at 192); by default for one-m-tile launches (decode
    // groups: the weights stream from HBM), CM_GEMM256_WST = 2 never, 3 always (A/B)
    static const int wst_env = getenv("CM_GEMM256_WST") ? atoi(getenv("CM_GEMM256_WST")) : 0;
    const bool w3 = split && (wst_env == 3 || (wst_env == 0 && a.M <= bm)) && a.K / TBK / a.ksplit >= 3;
#define CM_G256(SP, EP) do { if (SP == 2 && bm == 64) { if (bn == 256) launch_one<2, EP, 256, 3, 64>(a, blocks, s); else launch_one<2, EP, 192, 3, 64>(a, blocks, s); } \
                             else if (bn == 256) { if (SP == 2 && w3) launch_one<2, EP, 256, 3>(a, blocks, s); else launch_one<SP, EP, 256>(a, blocks, s); } \
                             else { if (SP == 2 && w3) launch_one<2, EP, 192, 3>(a, blocks, s); else launch_one<SP, EP, 192>(a, blocks, s); } } while (0)
#define CM_G256_EPI(SP) do { if (e == GEPI_STORE) CM_G256(SP, GEPI_STORE); else if (e == GEPI_RESADD) CM_G256(SP, GEPI_RESADD); \
        else if (e == GEPI_ACT_SPLIT) CM_G256(SP, GEPI_ACT_SPLIT); else if (e == GEPI_SILUMUL) CM_G256(SP, GEPI_SILUMUL); \
        else CM_G256(SP, GEPI_PARTIAL); } while (0)
    if (split) CM_G256_EPI(2); else CM_G256_EPI(1);
#undef CM_G256_EPI
#undef CM_G256
    return true;
}

}  // namespace cm
