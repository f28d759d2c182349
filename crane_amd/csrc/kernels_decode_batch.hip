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

constexpr int BKT = 2048;     // K tile staged in LDS (elements)

template <int PRO, int EPI, int MB>
__global__ __launch_bounds__(256, 2) void gemvb_kernel(GemvBArgs a) {
    extern __shared__ __attribute__((aligned(16))) float xs[];     // [MB][BKT] + scratch
    constexpr int R = 4, CU = 2;       // 4 rows x 2 chunks = 8 x 16-byte loads in flight per lane
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = a.K, N = a.N;
    const int nkt = (K + BKT - 1) / BKT;
    float* red = xs + MB * BKT;                                      // [4][MB] partial sums of squares
    const int r0 = (blockIdx.x * 4 + wave) * R;
    const uint16_t* wp[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int ri = (r0 + i < N) ? r0 + i : N - 1;
        wp[i] = a.W + (size_t)ri * a.ldw + lane * 8;
    }
    float acc[R][MB];
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
        for (int m = 0; m < MB; ++m) acc[i][m] = 0.f;
    float ss[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) ss[m] = 0.f;

    for (int kt = 0; kt < nkt; ++kt) {
        const int k0 = kt * BKT;
        const int tile = min(BKT, K - k0);            // multiple of 8
        __syncthreads();                              // previous tile fully consumed
        // ---- stage x[m][k0 .. k0+tile) for all m (zero-fill the ragged end) ----
        for (int e = tid; e < MB * (BKT / 4); e += 256) {
            const int m = e / (BKT / 4), k4 = e % (BKT / 4), k = k4 << 2;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (k < tile) {
                v = *(const f32x4*)(a.x + (size_t)m * a.ldx + k0 + k);
                if (PRO == PRO_RMSNORM) {
                    float s2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
#pragma unroll
                    for (int mm = 0; mm < MB; ++mm) if (mm == m) ss[mm] += s2;
                    const f32x4 w = *(const f32x4*)(a.nw + k0 + k);
                    v[0] *= w[0]; v[1] *= w[1]; v[2] *= w[2]; v[3] *= w[3];
                }
            }
            const int c = k >> 9, j = k & 511;
            ((f32x4*)(xs + m * BKT))[c * 128 + ((j >> 2) & 1) * 64 + (j >> 3)] = v;
        }
        __syncthreads();
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
                        const f32x4 xa = ((const f32x4*)(xs + m * BKT))[(c + u) * 128 + lane];
                        const f32x4 xb = ((const f32x4*)(xs + m * BKT))[(c + u) * 128 + 64 + lane];
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
    }
    // ---- RMSNorm scales ----
    float scale[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) scale[m] = 1.f;
    if (PRO == PRO_RMSNORM) {
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const float s2 = wave_sum(ss[m]);
            if (lane == 0) red[wave * MB + m] = s2;
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const float tot = (red[m] + red[MB + m]) + (red[2 * MB + m] + red[3 * MB + m]);
            scale[m] = 1.0f / sqrtf(tot / (float)K + a.eps);
        }
    }
    // ---- reduce + epilogue: lane (i * MB + m) keeps output (row r0 + i, sequence m) ----
    float mine = 0.f, mine_up = 0.f;
    float best[MB]; int besti[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) { best[m] = -INFINITY; besti[m] = 0x7FFFFFFF; }
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
    if (EPI == EPI_ARGMAX) {
        __syncthreads();
        int* redi = (int*)(red + 4 * MB);
        if (lane == 0) {
#pragma unroll
            for (int m = 0; m < MB; ++m) { red[wave * MB + m] = best[m]; redi[wave * MB + m] = besti[m]; }
        }
        __syncthreads();
        if (tid < MB && tid < a.n_seq) {
            float b = red[tid]; int bi = redi[tid];
            for (int w = 1; w < 4; ++w)
                if (red[w * MB + tid] > b || (red[w * MB + tid] == b && redi[w * MB + tid] < bi)) { b = red[w * MB + tid]; bi = redi[w * MB + tid]; }
            a.pmax[(size_t)tid * gridDim.x + blockIdx.x] = b;
            a.pidx[(size_t)tid * gridDim.x + blockIdx.x] = bi;
        }
    }
}

int gemvb_grid(int N) { return (N + 15) / 16; }

template <int PRO, int EPI>
static void launch_gemvb_t(const GemvBArgs& a, hipStream_t s) {
    const dim3 g(gemvb_grid(a.N)), b(256);
#define CM_GB(MBV) { const size_t lds = (size_t)MBV * BKT * 4 + 8 * MBV * 4 + 64; \
        hipLaunchKernelGGL((gemvb_kernel<PRO, EPI, MBV>), g, b, lds, s, a); }
    if (a.n_seq <= 2) CM_GB(2) else if (a.n_seq <= 4) CM_GB(4) else CM_GB(8)
#undef CM_GB
}

void launch_gemvb(int pro, int epi, const GemvBArgs& a, hipStream_t s) {
    if (pro == PRO_PLAIN) {
        if (epi == EPI_STORE) launch_gemvb_t<PRO_PLAIN, EPI_STORE>(a, s);
        else if (epi == EPI_RESADD) launch_gemvb_t<PRO_PLAIN, EPI_RESADD>(a, s);
        else if (epi == EPI_SILUMUL) launch_gemvb_t<PRO_PLAIN, EPI_SILUMUL>(a, s);
        else launch_gemvb_t<PRO_PLAIN, EPI_ARGMAX>(a, s);
    } else {
        if (epi == EPI_STORE) launch_gemvb_t<PRO_RMSNORM, EPI_STORE>(a, s);
        else if (epi == EPI_RESADD) launch_gemvb_t<PRO_RMSNORM, EPI_RESADD>(a, s);
        else if (epi == EPI_SILUMUL) launch_gemvb_t<PRO_RMSNORM, EPI_SILUMUL>(a, s);
        else launch_gemvb_t<PRO_RMSNORM, EPI_ARGMAX>(a, s);
    }
}

}  // namespace cm
