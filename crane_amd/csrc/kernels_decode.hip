// Decode-step (S = 1) kernels for gfx950.  HBM-bound: every weight byte is read once
// per token, so the design goal is "all 256 CUs streaming 16-byte non-temporal loads
// with >= 32 KiB in flight per CU", with everything else (RMSNorm, QK-norm, RoPE,
// KV append, SiLU*mul, residual add, arg-max) fused into a prologue or epilogue of
// the kernel that streams the weights.
//
// Replaces, per decode token, the ~15 candle launches per layer of
//   Qwen3Model::decode / DecoderLayer::forward / Attention::forward / Mlp::forward
//   (reference crane-core/src/models/qwen3/modeling.rs:307-533, 608-642, 698-716, 984-1036)
// with 6 launches per layer (see DESIGN.md "decode step").
#include <cstdio>
#include <cstdlib>

#include "dev_common.h"
#include "kernels.h"

namespace cm {

// =====================================================================================
// embedding row gather (modeling.rs:951): x[H] = f32(embed[token, :])
// =====================================================================================
__global__ void embed_row_kernel(const uint16_t* __restrict__ emb, const StepState* __restrict__ st,
                                 float* __restrict__ x, int H, int V) {
    const int bq = blockIdx.y;                                    // batched step: one row per sequence
    uint32_t tok = st[bq].token;
    if (tok >= (uint32_t)V) tok = 0;   // host validates ids; device stays in-bounds regardless
    int i = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i < H) {
        u32x2 p = *(const u32x2*)(emb + (size_t)tok * H + i);
        f32x4 o = {bf16_lo(p[0]), bf16_hi(p[0]), bf16_lo(p[1]), bf16_hi(p[1])};
        *(f32x4*)(x + (size_t)bq * H + i) = o;
    }
}

__global__ void set_state_kernel(StepState* st, uint32_t token, int32_t pos, int32_t slot, int32_t rope_delta) {
    st->token = token; st->pos = pos; st->slot = slot; st->rsv[0] = rope_delta;
}

// =====================================================================================
// fused GEMV:  y = epilogue( W[N,K](bf16) . prologue(x[K] f32) )
//   prologue PRO_RMSNORM: x <- x * nw / sqrt(mean(x^2)+eps)   (candle_nn::rms_norm, plain weight,
//                         modeling.rs:660-669,706,713,1024).  The 1/rms factor is applied to
//                         the finished dot product (it is a scalar), so x*nw is staged once.
//   epilogue EPI_RESADD : y[n] = res[n] + acc                 (modeling.rs:710,715)
//            EPI_SILUMUL: rows are interleaved gate_j, up_j -> y[j] = silu(g)*u
//                         (Mlp::forward modeling.rs:608-631; fused_silu_mul fused_ops.cu:119)
//            EPI_ARGMAX : y[n] = acc (logits) + per-block (max, lowest index) partial
//                         (gpu_argmax phase 1, fused_ops.cu:251)
// Layout: one wave owns R=2 consecutive rows at a time and sweeps K in 512-element
// chunks: lane l loads 16 B (8 bf16) at k = chunk*512 + l*8, fully coalesced 1 KiB per
// wave-instruction, non-temporal.  x lives in LDS as f32, permuted so that the two
// ds_read_b128 a lane needs per chunk are conflict-free (lane-linear 16-B slots).
// =====================================================================================
// NW = waves per workgroup: 4, or 5 where that makes the row groups divide evenly over the CUs (Qwen3.8-27B: 5120 rows = 2560
// two-row groups = 10 per CU; with 4-wave blocks the 640 blocks of down_proj -- 70 KB of x each, two per CU -- ran in 1.25 rounds).
template <int PRO, int EPI, int R, int U, bool PIPE, bool KGUARD, int NW = 4>
__global__ __launch_bounds__(NW * 64, NW == 4 ? (PIPE ? 3 : 4) : 2) void gemv_bf16_kernel(GemvArgs a) {
    constexpr int NT = NW * 64;
    extern __shared__ __attribute__((aligned(16))) float xs[];   // [Kpad] + [16] scratch
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = a.K, N = a.N;
    const int nch = (K + 511) >> 9;
    const int Kpad = nch << 9;
    float* red = xs + Kpad;

    // ---- this wave's work: row groups g = gw, gw+TW, ...; each group = nbpg batches of
    //      R rows x U chunks (R*U 16-byte loads per lane).  Batches are software-pipelined
    //      through two register buffers so HBM loads stay in flight across the x staging,
    //      the barrier, the FMAs and the reductions.
    const int G = (N + R - 1) / R;
    const int gw = blockIdx.x * NW + wave, TW = gridDim.x * NW;
    const int nbpg = nch / U;
    const int my_groups = (gw < G) ? (G - gw + TW - 1) / TW : 0;
    const int NB = my_groups * nbpg;

    u32x4 qa[R][U], qb[PIPE ? R : 1][PIPE ? U : 1];
    auto load_batch = [&](u32x4 (&q)[R][U], int g, int cb) {
        const int r0 = g * R;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int ri = (r0 + i < N) ? r0 + i : N - 1;            // clamp: loads stay in-bounds
            const uint16_t* wp = a.W + (size_t)ri * a.ldw + lane * 8 + (size_t)cb * 512;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (KGUARD) q[i][u] = ((cb + u) * 512 + lane * 8 < K) ? ld_nt16(wp + u * 512) : (u32x4){0, 0, 0, 0};
                else q[i][u] = ld_nt16(wp + u * 512);
            }
        }
    };
    if (NB > 0) load_batch(qa, gw, 0);        // weights first: they do not depend on x

    // ---- stage x (L2-resident, tiny) into LDS; fused RMSNorm statistics ----
    float ss = 0.f;
    auto stage_one = [&](int k4, const f32x4& vin, const f32x4& win) {
        f32x4 v = vin;
        if (PRO == PRO_RMSNORM) {
            ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
            v[0] *= win[0]; v[1] *= win[1]; v[2] *= win[2]; v[3] *= win[3];
        }
        const int k = k4 << 2;
        const int c = k >> 9, j = k & 511;
        ((f32x4*)xs)[c * 128 + ((j >> 2) & 1) * 64 + (j >> 3)] = v;
    };
    // PRO_ATTNCOMB: x is not materialised -- it is the merge of the NS per-head partials left by
    // attn_decode_head_kernel: x[h*D + d] = sum_s e^{m_s - M} o_s[d] / sum_s e^{m_s - M} l_s  (* sigmoid(gate))
    auto xload = [&](int k4i) -> f32x4 {
        if (PRO == PRO_GDNNORM) {
            // a value head = 128 consecutive values = the float4 of 32 consecutive lanes (an aligned half-wave: NT and K / 4 are
            // multiples of 32, so a half-wave is active as a whole): sum of squares over the head with five xor steps
            f32x4 v = *(const f32x4*)(a.x + (k4i << 2));
            float ss = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
            ss += __shfl_xor(ss, 1); ss += __shfl_xor(ss, 2); ss += __shfl_xor(ss, 4); ss += __shfl_xor(ss, 8); ss += __shfl_xor(ss, 16);
            const float rms = 1.0f / sqrtf(ss / 128.0f + a.eps);
            const f32x4 z = *(const f32x4*)(a.gdn_z + (k4i << 2));
            const f32x4 w = *(const f32x4*)(a.gdn_w + ((k4i & 31) << 2));
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] * rms * w[e] * (z[e] / (1.0f + expf(-z[e])));
            return v;
        }
        if (PRO != PRO_ATTNCOMB) return *(const f32x4*)(a.x + (k4i << 2));
        const int k = k4i << 2, h = k >> a.dshift, d = k & ((1 << a.dshift) - 1);
        const float* ml = a.part_ml + (size_t)h * a.ns * 2;
        const float* po = a.x + ((size_t)h * a.ns << a.dshift) + d;
        float M = -INFINITY;
        for (int s2 = 0; s2 < a.ns; ++s2) M = fmaxf(M, ml[2 * s2]);
        float Ls = 0.f;
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        for (int s2 = 0; s2 < a.ns; ++s2) {
            const float mm = ml[2 * s2];
            const float w = (mm > -INFINITY) ? expf(mm - M) : 0.f;
            Ls += w * ml[2 * s2 + 1];
            const f32x4 p = *(const f32x4*)(po + ((size_t)s2 << a.dshift));
            o[0] += w * p[0]; o[1] += w * p[1]; o[2] += w * p[2]; o[3] += w * p[3];
        }
        const float inv = 1.0f / Ls;
        o[0] *= inv; o[1] *= inv; o[2] *= inv; o[3] *= inv;
        if (a.gate != nullptr) {
            const f32x4 g = *(const f32x4*)(a.gate + k);
            o[0] *= 1.0f / (1.0f + expf(-g[0])); o[1] *= 1.0f / (1.0f + expf(-g[1]));
            o[2] *= 1.0f / (1.0f + expf(-g[2])); o[3] *= 1.0f / (1.0f + expf(-g[3]));
        }
        return o;
    };
    const int n4 = K >> 2;                       // K % 8 == 0
    int k4 = tid;
    for (; k4 + 3 * NT < n4; k4 += 4 * NT) {     // 4 independent loads in flight per thread
        f32x4 v[4]; f32x4 w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = xload(k4 + i * NT);
            if (PRO == PRO_RMSNORM) w[i] = *(const f32x4*)(a.nw + ((k4 + i * NT) << 2));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) stage_one(k4 + i * NT, v[i], w[i]);
    }
    for (; k4 < n4; k4 += NT) {
        f32x4 v = xload(k4);
        f32x4 w = {0.f, 0.f, 0.f, 0.f};
        if (PRO == PRO_RMSNORM) w = *(const f32x4*)(a.nw + (k4 << 2));
        stage_one(k4, v, w);
    }
    if (KGUARD) {                                 // zero the K..Kpad tail
        for (int z = n4 + tid; z < (Kpad >> 2); z += NT) {
            const int k = z << 2, c = k >> 9, j = k & 511;
            ((f32x4*)xs)[c * 128 + ((j >> 2) & 1) * 64 + (j >> 3)] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    float scale = 1.f;
    if (PRO == PRO_RMSNORM) {
        ss = wave_sum(ss);
        if (lane == 0) red[wave] = ss;
    }
    __syncthreads();
    if (PRO == PRO_RMSNORM) {
        float tot = (red[0] + red[1]) + (red[2] + red[3]);
        if (NW > 4) { for (int w2 = 4; w2 < NW; ++w2) tot += red[w2]; }
        scale = 1.0f / sqrtf(tot / (float)K + a.eps);
    }

    float best = -INFINITY; int besti = 0x7FFFFFFF;
    float acc[R];
#pragma unroll
    for (int i = 0; i < R; ++i) acc[i] = 0.f;
    int g = gw, bi = 0;                          // current group, batch index inside it

    auto compute = [&](u32x4 (&q)[R][U]) {
        const int cb = bi * U;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const f32x4 xa = ((const f32x4*)xs)[(cb + u) * 128 + lane];
            const f32x4 xb = ((const f32x4*)xs)[(cb + u) * 128 + 64 + lane];
#pragma unroll
            for (int i = 0; i < R; ++i) {
                acc[i] += bf16_lo(q[i][u][0]) * xa[0] + bf16_hi(q[i][u][0]) * xa[1] +
                          bf16_lo(q[i][u][1]) * xa[2] + bf16_hi(q[i][u][1]) * xa[3] +
                          bf16_lo(q[i][u][2]) * xb[0] + bf16_hi(q[i][u][2]) * xb[1] +
                          bf16_lo(q[i][u][3]) * xb[2] + bf16_hi(q[i][u][3]) * xb[3];
            }
        }
        if (++bi < nbpg) return;
        // ---- group finished: reduce + epilogue ----
        const int r0 = g * R;
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
            if (lane < R && r0 + lane < N) {
                // in-place residual (the decoder's use): a fire-and-forget f32 atomic is the same single add and, unlike
                // load + store, does not make the wave drain the weight loads it has in flight
                if (PIPE && a.res == a.y) atomicAdd(&a.y[r0 + lane], mine);
                else a.y[r0 + lane] = a.res[r0 + lane] + mine;
            }
        } else if (EPI == EPI_SILUMUL) {
            if (lane < R / 2 && r0 + 2 * lane + 1 < N)
                a.y[(r0 >> 1) + lane] = (mine / (1.0f + expf(-mine))) * mine_up;
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
        bi = 0;
        g += TW;
    };
    // (group, chunk) of flattened batch index b
    auto batch_g = [&](int b) { return gw + (b / nbpg) * TW; };
    auto batch_cb = [&](int b) { return (b % nbpg) * U; };

    if constexpr (PIPE) {
        // Two statically named register sets, and NO load under a branch inside the steady state: a guarded prefetch
        // makes the compiler wait with vmcnt(0) at the next use, which also waits for the batch just requested
        // (DESIGN 3.13).  The last one or two batches are peeled into straight-line code.
        int b = 0;
        while (b + 2 < NB) {
            load_batch(qb, batch_g(b + 1), batch_cb(b + 1));
            compute(qa);
            load_batch(qa, batch_g(b + 2), batch_cb(b + 2));
            compute(qb);
            b += 2;
        }
        if (b + 1 < NB) {
            load_batch(qb, batch_g(b + 1), batch_cb(b + 1));
            compute(qa);
            compute(qb);
        } else if (b < NB) {
            compute(qa);
        }
    } else {
        for (int b = 0; b < NB; ++b) {
            compute(qa);
            if (b + 1 < NB) load_batch(qa, batch_g(b + 1), batch_cb(b + 1));
        }
    }
    if (EPI == EPI_ARGMAX) {
        __syncthreads();
        int* redi = (int*)(red + 8);
        if (lane == 0) { red[wave] = best; redi[wave] = besti; }
        __syncthreads();
        if (tid == 0) {
            float bb = red[0]; int bbi = redi[0];
            for (int w = 1; w < NW; ++w)
                if (red[w] > bb || (red[w] == bb && redi[w] < bbi)) { bb = red[w]; bbi = redi[w]; }
            a.pmax[blockIdx.x] = bb; a.pidx[blockIdx.x] = bbi;
        }
    }
}

// arg-max phase 2 (fused_ops.cu:322) + device-side advance of the autoregressive state:
// the next graph replay consumes st->token / st->pos without a host round trip.
__global__ __launch_bounds__(256) void argmax_final_kernel(const float* __restrict__ pmax,
                                                           const int* __restrict__ pidx, int n,
                                                           StepState* st, uint32_t* ring, int ring_mask,
                                                           int advance, int slabs, size_t slab_stride) {
    __shared__ float sm[256];
    __shared__ int si[256];
    pmax += (size_t)blockIdx.x * n; pidx += (size_t)blockIdx.x * n; st += blockIdx.x;   // batched step: one block per sequence
    float b = -INFINITY; int bi = 0x7FFFFFFF;
    for (int sl = 0; sl < slabs; ++sl)            // batched step under TP: one [n_seq][n] slab per rank (vocabulary shard)
        for (int i = threadIdx.x; i < n; i += 256) {
            float v = pmax[sl * slab_stride + i]; int ix = pidx[sl * slab_stride + i];
            // (a partial of a wave that produced no row, or saw only NaN / -inf logits, carries no candidate: its index is a sentinel
            // -- plus a shard's base, possibly wrapped negative -- and must not win a tie at -inf)
            if (!(v > -INFINITY) || ix < 0) continue;
            if (v > b || (v == b && ix < bi)) { b = v; bi = ix; }
        }
    sm[threadIdx.x] = b; si[threadIdx.x] = bi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            float v = sm[threadIdx.x + s]; int ix = si[threadIdx.x + s];
            if (v > sm[threadIdx.x] || (v == sm[threadIdx.x] && ix < si[threadIdx.x])) {
                sm[threadIdx.x] = v; si[threadIdx.x] = ix;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        uint32_t t = si[0] == 0x7FFFFFFF ? 0u : (uint32_t)si[0];      // no finite logit anywhere: token 0 instead of an out-of-range id
        st->next = t;
        if (advance) {
            ring[st->pad & ring_mask] = t;
            st->pad += 1;
            st->token = t;
            st->pos += 1;
        }
    }
}

// =====================================================================================
// host-side launchers
// =====================================================================================
// (rows per group, chunks per batch, software-pipelined) per K-shape class; overridable for
// tuning with CM_GEMV_CFG="R,U,P" (must divide the chunk count).
struct GemvCfg { int R, U, P; };
static GemvCfg gemv_cfg(int K) {
    const int nch = (K + 511) / 512;
    if (K % 512 != 0) return {2, 1, 0};
    static int eR = -1, eU = 0, eP = 0;
    if (eR == -1) {
        eR = 0;
        if (const char* e = getenv("CM_GEMV_CFG")) { if (sscanf(e, "%d,%d,%d", &eR, &eU, &eP) != 3) eR = 0; }
    }
    if (eR > 0 && nch % eU == 0) return {eR, eU, eP};
    if (nch % 4 == 0) return {2, 4, 0};      // measured best on MI355X (profiles/r01_gemv_variant_sweep.log)
    if (nch % 2 == 0) return {2, 2, 1};      // K = 5120 / 17408 (Qwen3.8-27B): 106.2 tok/s vs 102.0 with {8, 2, 0}
    // odd chunk counts (K = 3584: Qwen3.5-0.8B down_proj): the whole row pair in one batch keeps 2 x nch loads in flight per
    // wave AND two-row groups keep the wave count up for small N ({8, 1, 0} left 32 blocks for N = 1024: 0.8 TB/s)
    if (nch == 1 || nch == 3 || nch == 5 || nch == 7) return {2, nch, 0};
    return {2, 1, 1};
}

template <int PRO, int EPI>
static void launch_gemv_t(const GemvArgs& a, int grid, hipStream_t s) {
    const int nch = (a.K + 511) / 512;
    const size_t lds = (size_t)nch * 512 * 4 + 64;
    const dim3 g(grid < 0 ? -grid : grid), b(256);
    const GemvCfg c = gemv_cfg(a.K);
    if (grid < 0) {            // five-wave workgroups (gemv_grid's encoding): |grid| blocks of 320 threads, two per CU
        if (c.R == 2 && c.U == 2 && c.P == 1) { hipLaunchKernelGGL((gemv_bf16_kernel<PRO, EPI, 2, 2, true, false, 5>), dim3(-grid), dim3(320), lds, s, a); return; }
        if (c.R == 2 && c.U == 4 && c.P == 0) { hipLaunchKernelGGL((gemv_bf16_kernel<PRO, EPI, 2, 4, false, false, 5>), dim3(-grid), dim3(320), lds, s, a); return; }
    }
#define CM_GEMV_CASE(RR, UU, PP) \
    if (c.R == RR && c.U == UU && c.P == PP) { hipLaunchKernelGGL((gemv_bf16_kernel<PRO, EPI, RR, UU, PP != 0, false>), g, b, lds, s, a); return; }
    if (a.K % 512 != 0) { hipLaunchKernelGGL((gemv_bf16_kernel<PRO, EPI, 2, 1, false, true>), g, b, lds, s, a); return; }
    CM_GEMV_CASE(2, 8, 0) CM_GEMV_CASE(2, 8, 1) CM_GEMV_CASE(2, 4, 0) CM_GEMV_CASE(2, 4, 1)
    CM_GEMV_CASE(4, 4, 0) CM_GEMV_CASE(4, 4, 1) CM_GEMV_CASE(4, 2, 0) CM_GEMV_CASE(4, 2, 1)
    CM_GEMV_CASE(8, 2, 0) CM_GEMV_CASE(8, 2, 1) CM_GEMV_CASE(8, 1, 0) CM_GEMV_CASE(8, 1, 1)
    CM_GEMV_CASE(2, 2, 1)
    CM_GEMV_CASE(2, 1, 0) CM_GEMV_CASE(2, 3, 0) CM_GEMV_CASE(2, 5, 0) CM_GEMV_CASE(2, 7, 0) CM_GEMV_CASE(2, 1, 1)
#undef CM_GEMV_CASE
    hipLaunchKernelGGL((gemv_bf16_kernel<PRO, EPI, 8, 1, false, false>), g, b, lds, s, a);
}

int gemv_rows_per_group(int K) { return gemv_cfg(K).R; }

int gemv_grid(int N, int K, int num_cu, bool allow_nw5) {
    // one wave per row group until the chip is full, then grid-stride
    const GemvCfg c = gemv_cfg(K);
    const int groups = (N + c.R - 1) / c.R;
    // five-wave workgroups, returned as a NEGATIVE block count: when the groups are exactly 5 or 10 per CU (one wave each, no
    // grid stride, every CU the same share) and the four-wave grid would not be co-resident or even
    static int nw5 = -1;
    if (nw5 < 0) nw5 = getenv("CM_GEMV_NW5") ? atoi(getenv("CM_GEMV_NW5")) : 1;
    if (nw5 && allow_nw5 && K % 512 == 0 && ((c.R == 2 && c.U == 2 && c.P == 1) || (c.R == 2 && c.U == 4 && c.P == 0)) && N % c.R == 0 &&
        groups % 5 == 0 && (groups / 5 == num_cu || groups / 5 == 2 * num_cu) &&
        (size_t)(groups / 5 / num_cu) * ((size_t)((K + 511) / 512) * 2048 + 64) <= 160 * 1024 - 1024)
        return -(groups / 5);
    int blocks = (groups + 3) / 4;
    static int per_cu = -1;
    if (per_cu < 0) { per_cu = 0; if (const char* e = getenv("CM_GEMV_BLOCKS_PER_CU")) per_cu = atoi(e); }
    const int cap = num_cu * (per_cu > 0 ? per_cu : (c.P ? 3 : 4));
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return blocks;
}

void launch_gemv(int pro, int epi, const GemvArgs& a, int grid, hipStream_t s) {
    if (pro == PRO_ATTNCOMB) {          // only o_proj uses it (row-parallel under TP: store, else residual add)
        if (epi == EPI_STORE) launch_gemv_t<PRO_ATTNCOMB, EPI_STORE>(a, grid, s);
        else launch_gemv_t<PRO_ATTNCOMB, EPI_RESADD>(a, grid, s);
        return;
    }
    if (pro == PRO_GDNNORM) {           // out_proj of a Gated-Delta-Net layer (row-parallel under TP: store, else residual add)
        if (epi == EPI_STORE) launch_gemv_t<PRO_GDNNORM, EPI_STORE>(a, grid, s);
        else launch_gemv_t<PRO_GDNNORM, EPI_RESADD>(a, grid, s);
        return;
    }
    if (pro == PRO_PLAIN) {
        if (epi == EPI_STORE) launch_gemv_t<PRO_PLAIN, EPI_STORE>(a, grid, s);
        else if (epi == EPI_RESADD) launch_gemv_t<PRO_PLAIN, EPI_RESADD>(a, grid, s);
        else if (epi == EPI_SILUMUL) launch_gemv_t<PRO_PLAIN, EPI_SILUMUL>(a, grid, s);
        else launch_gemv_t<PRO_PLAIN, EPI_ARGMAX>(a, grid, s);
    } else {
        if (epi == EPI_STORE) launch_gemv_t<PRO_RMSNORM, EPI_STORE>(a, grid, s);
        else if (epi == EPI_RESADD) launch_gemv_t<PRO_RMSNORM, EPI_RESADD>(a, grid, s);
        else if (epi == EPI_SILUMUL) launch_gemv_t<PRO_RMSNORM, EPI_SILUMUL>(a, grid, s);
        else launch_gemv_t<PRO_RMSNORM, EPI_ARGMAX>(a, grid, s);
    }
}

void launch_embed_row(const uint16_t* emb, const StepState* st, float* x, int H, int V, int n_seq, hipStream_t s) {
    hipLaunchKernelGGL(embed_row_kernel, dim3((H / 4 + 255) / 256, n_seq), dim3(256), 0, s, emb, st, x, H, V);
}
void launch_set_state(StepState* st, uint32_t token, int32_t pos, int32_t slot, int32_t rope_delta, hipStream_t s) {
    hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(1), 0, s, st, token, pos, slot, rope_delta);
}
void launch_argmax_final(const float* pmax, const int* pidx, int n, StepState* st, uint32_t* ring,
                         int ring_mask, int advance, int n_seq, hipStream_t s, int slabs, size_t slab_stride) {
    hipLaunchKernelGGL(argmax_final_kernel, dim3(n_seq), dim3(256), 0, s, pmax, pidx, n, st, ring, ring_mask, advance, slabs, slab_stride);
}

}  // namespace cm
