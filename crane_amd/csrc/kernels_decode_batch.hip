// Batched decode (continuous batching, <= 8 sequences per step): the weight stream is read ONCE for
// all sequences.  Replaces step_batch_decode + pad_and_stack_kv_caches + extract_batch_kv
// (reference qwen3/modeling.rs:1141-1378, crane-serve/src/engine/mod.rs:822-1057): no padding, no KV
// copies -- every sequence keeps its own page table and position.
//
// gemvb: y[m, n] = epilogue( W[n, :] . prologue(x[m, :]) ) for m < MB.  A wave owns 8 consecutive rows
// (8 x 16-byte non-temporal loads in flight per lane per 512-element chunk); x lives in LDS as f32 in
// K-tiles of 2048 elements ([MB][2048] = 64 KiB at MB = 8) with the same conflict-free permutation as
// the single-sequence GEMV; 8 x MB accumulators per lane.  Same fused prologue/epilogues as gemv.
#include "dev_common.h"
#include "kernels.h"

namespace cm {

constexpr int BKT = 4096;     // K tile staged in LDS (elements): [MB][4096] f32 = 128 KiB at MB = 8
constexpr int BW = 8;         // waves per block

// K <= BKT : x is staged ONCE per block, waves grid-stride over row groups (staging traffic << weight traffic).
// K >  BKT : one row group per wave, x restaged per K tile, accumulators carried across tiles.
template <int PRO, int EPI, int MB>
__global__ __launch_bounds__(512, 2) void gemvb_kernel(GemvBArgs a) {
    extern __shared__ __attribute__((aligned(16))) float xs[];     // [MB][tk] + scratch
    constexpr int R = 4, CU = 2;       // 4 rows x 2 chunks = 8 x 16-byte loads in flight per lane
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = a.K, N = a.N;
    const int tk = ((min(K, BKT) + 511) >> 9) << 9;                  // LDS row length (elements)
    const int nkt = (K + BKT - 1) / BKT;
    float* red = xs + MB * tk;                                       // [BW][MB] + arg-max scratch
    const int G = (N + R - 1) / R;
    float ss[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) ss[m] = 0.f;
    float scale[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) scale[m] = 1.f;
    float best[MB]; int besti[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) { best[m] = -INFINITY; besti[m] = 0x7FFFFFFF; }

    auto stage = [&](int kt, bool count) {
        const int k0 = kt * BKT;
        const int tile = min(BKT, K - k0);
        // 4 independent 16-byte loads per thread per batch (one L2 round trip per batch, not per element)
        const int total = MB * (tk / 4);
        for (int e0 = tid; e0 < total; e0 += 64 * BW * 4) {
            f32x4 vv[4], ww[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = e0 + i * 64 * BW;
                const int m = e / (tk / 4), k = (e % (tk / 4)) << 2;
                const bool live = e < total && k < tile;
                if (PRO == PRO_ATTNCOMB) {            // merge the per-head partials of attn_decode_head_kernel
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (live) {
                        const int kk = k0 + k, h = kk >> a.dshift, d = kk & ((1 << a.dshift) - 1);
                        const float* ml = a.part_ml + ((size_t)m * (a.K >> a.dshift) + h) * a.ns * 2;
                        const float* po = a.x + (size_t)m * a.ldx + ((size_t)h * a.ns << a.dshift) + d;
                        float M = -INFINITY;
                        for (int s2 = 0; s2 < a.ns; ++s2) M = fmaxf(M, ml[2 * s2]);
                        float Ls = 0.f;
                        for (int s2 = 0; s2 < a.ns; ++s2) {
                            const float mm = ml[2 * s2];
                            const float w = (mm > -INFINITY) ? expf(mm - M) : 0.f;
                            Ls += w * ml[2 * s2 + 1];
                            const f32x4 p = *(const f32x4*)(po + ((size_t)s2 << a.dshift));
                            v[0] += w * p[0]; v[1] += w * p[1]; v[2] += w * p[2]; v[3] += w * p[3];
                        }
                        const float inv = 1.0f / Ls;
                        v[0] *= inv; v[1] *= inv; v[2] *= inv; v[3] *= inv;
                        if (a.gate != nullptr) {
                            const f32x4 g = *(const f32x4*)(a.gate + (size_t)m * a.gate_stride + kk);
                            v[0] *= 1.0f / (1.0f + expf(-g[0])); v[1] *= 1.0f / (1.0f + expf(-g[1]));
                            v[2] *= 1.0f / (1.0f + expf(-g[2])); v[3] *= 1.0f / (1.0f + expf(-g[3]));
                        }
                    }
                    vv[i] = v;
                } else
                vv[i] = live ? *(const f32x4*)(a.x + (size_t)m * a.ldx + k0 + k) : (f32x4){0.f, 0.f, 0.f, 0.f};
                if (PRO == PRO_RMSNORM) ww[i] = live ? *(const f32x4*)(a.nw + k0 + k) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = e0 + i * 64 * BW;
                if (e >= total) continue;
                const int m = e / (tk / 4), k = (e % (tk / 4)) << 2;
                f32x4 v = vv[i];
                if (PRO == PRO_RMSNORM) {
                    if (count) {
                        const float s2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
#pragma unroll
                        for (int mm = 0; mm < MB; ++mm) if (mm == m) ss[mm] += s2;
                    }
                    v[0] *= ww[i][0]; v[1] *= ww[i][1]; v[2] *= ww[i][2]; v[3] *= ww[i][3];
                }
                const int c = k >> 9, j = k & 511;
                ((f32x4*)(xs + m * tk))[c * 128 + ((j >> 2) & 1) * 64 + (j >> 3)] = v;
            }
        }
    };
    auto finish_scale = [&]() {       // block-wide sum of squares -> 1/rms per sequence
        if (PRO == PRO_RMSNORM) {
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                const float s2 = wave_sum(ss[m]);
                if (lane == 0) red[wave * MB + m] = s2;
            }
            __syncthreads();
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                float tot = 0.f;
#pragma unroll
                for (int w = 0; w < BW; ++w) tot += red[w * MB + m];
                scale[m] = 1.0f / sqrtf(tot / (float)K + a.eps);
            }
            __syncthreads();
        }
    };

    float acc[R][MB];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
            for (int m = 0; m < MB; ++m) acc[i][m] = 0.f;
    };
    auto sweep = [&](int g, int kt) {           // accumulate tile kt of row group g
        const int r0 = g * R;
        const uint16_t* wp[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int ri = (r0 + i < N) ? r0 + i : N - 1;
            wp[i] = a.W + (size_t)ri * a.ldw + lane * 8;
        }
        const int tile = min(BKT, K - kt * BKT);
        const int tchunks = (tile + 511) >> 9;
        for (int c = 0; c < tchunks; c += CU) {
            u32x4 q[R][CU];
#pragma unroll
            for (int u = 0; u < CU; ++u) {
                const int cg = kt * (BKT / 512) + c + u;
                const bool ok = (c + u < tchunks) && (cg * 512 + lane * 8) < K;     // ragged tail of K
#pragma unroll
                for (int i = 0; i < R; ++i) q[i][u] = ok ? ld_nt16(wp[i] + (size_t)cg * 512) : (u32x4){0, 0, 0, 0};
            }
#pragma unroll
            for (int u = 0; u < CU; ++u) {
                if (c + u < tchunks) {
#pragma unroll
                    for (int m = 0; m < MB; ++m) {
                        const f32x4 xa = ((const f32x4*)(xs + m * tk))[(c + u) * 128 + lane];
                        const f32x4 xb = ((const f32x4*)(xs + m * tk))[(c + u) * 128 + 64 + lane];
#pragma unroll
                        for (int i = 0; i < R; ++i) {
                            acc[i][m] += bf16_lo(q[i][u][0]) * xa[0] + bf16_hi(q[i][u][0]) * xa[1] + bf16_lo(q[i][u][1]) * xa[2] +
                                         bf16_hi(q[i][u][1]) * xa[3] + bf16_lo(q[i][u][2]) * xb[0] + bf16_hi(q[i][u][2]) * xb[1] +
                                         bf16_lo(q[i][u][3]) * xb[2] + bf16_hi(q[i][u][3]) * xb[3];
                        }
                    }
                }
            }
        }
    };
    auto epilogue = [&](int g) {                // reduce + store: lane (i * MB + m) keeps (row r0 + i, sequence m)
        const int r0 = g * R;
        float mine = 0.f, mine_up = 0.f;
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                acc[i][m] = wave_sum(acc[i][m]) * scale[m];
                if (EPI != EPI_SILUMUL && lane == i * MB + m) mine = acc[i][m];
                if (EPI == EPI_SILUMUL && (i & 1) && lane == (i >> 1) * MB + m) { mine = acc[i - 1][m]; mine_up = acc[i][m]; }
                if (EPI == EPI_ARGMAX && r0 + i < N) {
                    const int ix = r0 + i + a.idx_base;
                    if (acc[i][m] > best[m] || (acc[i][m] == best[m] && ix < besti[m])) { best[m] = acc[i][m]; besti[m] = ix; }
                }
            }
        if (EPI == EPI_SILUMUL) {
            if (lane < (R / 2) * MB) {
                const int j = lane / MB, m = lane % MB;
                if (r0 + 2 * j + 1 < N && m < a.n_seq)
                    a.y[(size_t)m * a.ldy + (r0 >> 1) + j] = (mine / (1.0f + expf(-mine))) * mine_up;
            }
        } else if (lane < R * MB) {
            const int i = lane / MB, m = lane % MB;
            if (r0 + i < N && m < a.n_seq) {
                const size_t o = (size_t)m * a.ldy + r0 + i;
                a.y[o] = (EPI == EPI_RESADD) ? a.res[o] + mine : mine;
            }
        }
    };

    if (nkt == 1) {
        stage(0, true);
        __syncthreads();
        finish_scale();
        for (int g = blockIdx.x * BW + wave; g < G; g += gridDim.x * BW) {
            zero_acc();
            sweep(g, 0);
            epilogue(g);
        }
    } else {
        const int g = blockIdx.x * BW + wave;    // one group per wave (grid = ceil(G / BW))
        zero_acc();
        for (int kt = 0; kt < nkt; ++kt) {
            __syncthreads();                      // previous tile fully consumed
            stage(kt, true);
            __syncthreads();
            if (g < G) sweep(g, kt);
        }
        __syncthreads();
        finish_scale();
        if (g < G) epilogue(g);
    }
    if (EPI == EPI_ARGMAX) {
        __syncthreads();
        int* redi = (int*)(red + BW * MB);
        if (lane == 0) {
#pragma unroll
            for (int m = 0; m < MB; ++m) { red[wave * MB + m] = best[m]; redi[wave * MB + m] = besti[m]; }
        }
        __syncthreads();
        if (tid < MB && tid < a.n_seq) {
            float b = red[tid]; int bi = redi[tid];
            for (int w = 1; w < BW; ++w)
                if (red[w * MB + tid] > b || (red[w * MB + tid] == b && redi[w * MB + tid] < bi)) { b = red[w * MB + tid]; bi = redi[w * MB + tid]; }
            a.pmax[(size_t)tid * gridDim.x + blockIdx.x] = b;
            a.pidx[(size_t)tid * gridDim.x + blockIdx.x] = bi;
        }
    }
}

int gemvb_grid(int N, int K, int num_cu) {
    const int G = (N + 3) / 4, per_block = (G + BW - 1) / BW;
    if (K <= BKT) return std::max(1, std::min(per_block, num_cu));
    return std::max(1, per_block);
}

template <int PRO, int EPI>
static void launch_gemvb_t(const GemvBArgs& a, int grid, hipStream_t s) {
    const dim3 g(grid), b(64 * BW);
    const int tk = ((std::min(a.K, BKT) + 511) / 512) * 512;
#define CM_GB(MBV) { const size_t lds = (size_t)MBV * tk * 4 + 2 * BW * MBV * 4 + 64; \
        static DevOnce attr_##MBV; \
        attr_##MBV.run([] { (void)hipFuncSetAttribute((const void*)gemvb_kernel<PRO, EPI, MBV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }); \
        hipLaunchKernelGGL((gemvb_kernel<PRO, EPI, MBV>), g, b, lds, s, a); }
    if (a.n_seq <= 2) CM_GB(2) else if (a.n_seq <= 4) CM_GB(4) else if (a.n_seq <= 8) CM_GB(8)
    else {
        // more rows than the kernel keeps in LDS (the caller formed a group of up to 128 sequences for the matrix-core kernels and
        // this projection -- SiLU*mul or arg-max with K > 4096, Qwen3.8-27B / Qwen3-14B -- has no matrix-core form): passes of 8 rows,
        // every per-sequence array advanced by the pass (pmax / pidx rows are `grid` entries long)
        for (int m0 = 0; m0 < a.n_seq; m0 += 8) {
            GemvBArgs b8 = a;
            b8.n_seq = std::min(8, a.n_seq - m0);
            b8.x = a.x + (size_t)m0 * a.ldx;
            if (a.y) b8.y = a.y + (size_t)m0 * a.ldy;
            if (a.res) b8.res = a.res + (size_t)m0 * a.ldy;
            if (a.pmax) b8.pmax = a.pmax + (size_t)m0 * grid;
            if (a.pidx) b8.pidx = a.pidx + (size_t)m0 * grid;
            if (a.part_ml) b8.part_ml = a.part_ml + (size_t)m0 * (a.K >> a.dshift) * a.ns * 2;
            if (a.gate) b8.gate = a.gate + (size_t)m0 * a.gate_stride;
            launch_gemvb_t<PRO, EPI>(b8, grid, s);
        }
    }
#undef CM_GB
}

// grid (nblk, rows): block (b, r) scans its slice of row r -- strict > and the lowest index on ties, like every arg-max level
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ logits, int ld, int n, int idx_base,
                                                          float* __restrict__ pmax, int* __restrict__ pidx) {
    __shared__ float sm[256];
    __shared__ int si[256];
    const int row = blockIdx.y, nblk = gridDim.x;
    const int per = ((n + nblk - 1) / nblk + 3) & ~3;                    // columns per block, a multiple of 4 (rows are 16-byte aligned)
    const int c0 = blockIdx.x * per, c1 = min(n, c0 + per);
    const float* r = logits + (size_t)row * ld;
    float b = -INFINITY; int bi = 0x7FFFFFFF;
    for (int c = c0 + threadIdx.x * 4; c < c1; c += 1024) {
        if (c + 4 <= c1) {
            const f32x4 v = *(const f32x4*)(r + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) if (v[e] > b) { b = v[e]; bi = c + e; }
        } else {
            for (int e = 0; c + e < c1; ++e) if (r[c + e] > b) { b = r[c + e]; bi = c + e; }
        }
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
    if (threadIdx.x == 0) {
        pmax[(size_t)row * nblk + blockIdx.x] = sm[0];
        pidx[(size_t)row * nblk + blockIdx.x] = si[0] == 0x7FFFFFFF ? 0x7FFFFFFF : idx_base + si[0];
    }
}
void launch_argmax_rows(const float* logits, int ld, int n, int idx_base, float* pmax, int* pidx, int nblk, int n_rows, hipStream_t s) {
    hipLaunchKernelGGL(argmax_rows_kernel, dim3(nblk, n_rows), dim3(256), 0, s, logits, ld, n, idx_base, pmax, pidx);
}

void launch_gemvb(int pro, int epi, const GemvBArgs& a, int grid, hipStream_t s) {
    if (pro == PRO_ATTNCOMB) {
        if (epi == EPI_STORE) launch_gemvb_t<PRO_ATTNCOMB, EPI_STORE>(a, grid, s);
        else launch_gemvb_t<PRO_ATTNCOMB, EPI_RESADD>(a, grid, s);
        return;
    }
    if (pro == PRO_PLAIN) {
        if (epi == EPI_STORE) launch_gemvb_t<PRO_PLAIN, EPI_STORE>(a, grid, s);
        else if (epi == EPI_RESADD) launch_gemvb_t<PRO_PLAIN, EPI_RESADD>(a, grid, s);
        else if (epi == EPI_SILUMUL) launch_gemvb_t<PRO_PLAIN, EPI_SILUMUL>(a, grid, s);
        else launch_gemvb_t<PRO_PLAIN, EPI_ARGMAX>(a, grid, s);
    } else {
        if (epi == EPI_STORE) launch_gemvb_t<PRO_RMSNORM, EPI_STORE>(a, grid, s);
        else if (epi == EPI_RESADD) launch_gemvb_t<PRO_RMSNORM, EPI_RESADD>(a, grid, s);
        else if (epi == EPI_SILUMUL) launch_gemvb_t<PRO_RMSNORM, EPI_SILUMUL>(a, grid, s);
        else launch_gemvb_t<PRO_RMSNORM, EPI_ARGMAX>(a, grid, s);
    }
}

}  // namespace cm
