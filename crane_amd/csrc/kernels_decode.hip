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
    uint32_t tok = st->token;
    if (tok >= (uint32_t)V) tok = 0;   // host validates ids; device stays in-bounds regardless
    int i = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i < H) {
        u32x2 p = *(const u32x2*)(emb + (size_t)tok * H + i);
        f32x4 o = {bf16_lo(p[0]), bf16_hi(p[0]), bf16_lo(p[1]), bf16_hi(p[1])};
        *(f32x4*)(x + i) = o;
    }
}

__global__ void set_state_kernel(StepState* st, uint32_t token, int32_t pos) {
    st->token = token; st->pos = pos;
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
template <int PRO, int EPI, int R, int U, bool PIPE, bool KGUARD>
__global__ __launch_bounds__(256, PIPE ? 3 : 4) void gemv_bf16_kernel(GemvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float xs[];   // [Kpad] + [8] scratch
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
    const int gw = blockIdx.x * 4 + wave, TW = gridDim.x * 4;
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
    auto stage_one = [&](int k4, const f32x4& vin, const u32x2& win) {
        f32x4 v = vin;
        if (PRO == PRO_RMSNORM) {
            ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
            v[0] *= bf16_lo(win[0]); v[1] *= bf16_hi(win[0]);
            v[2] *= bf16_lo(win[1]); v[3] *= bf16_hi(win[1]);
        }
        const int k = k4 << 2;
        const int c = k >> 9, j = k & 511;
        ((f32x4*)xs)[c * 128 + ((j >> 2) & 1) * 64 + (j >> 3)] = v;
    };
    const int n4 = K >> 2;                       // K % 8 == 0
    int k4 = tid;
    for (; k4 + 768 < n4; k4 += 1024) {          // 4 independent loads in flight per thread
        f32x4 v[4]; u32x2 w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = *(const f32x4*)(a.x + ((k4 + i * 256) << 2));
            if (PRO == PRO_RMSNORM) w[i] = *(const u32x2*)(a.nw + ((k4 + i * 256) << 2));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) stage_one(k4 + i * 256, v[i], w[i]);
    }
    for (; k4 < n4; k4 += 256) {
        f32x4 v = *(const f32x4*)(a.x + (k4 << 2));
        u32x2 w = {0, 0};
        if (PRO == PRO_RMSNORM) w = *(const u32x2*)(a.nw + (k4 << 2));
        stage_one(k4, v, w);
    }
    if (KGUARD) {                                 // zero the K..Kpad tail
        for (int z = n4 + tid; z < (Kpad >> 2); z += 256) {
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
            if (lane < R && r0 + lane < N) a.y[r0 + lane] = a.res[r0 + lane] + mine;
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
        for (int b = 0; b < NB; b += 2) {
            if (b + 1 < NB) load_batch(qb, batch_g(b + 1), batch_cb(b + 1));
            compute(qa);
            if (b + 1 < NB) {
                if (b + 2 < NB) load_batch(qa, batch_g(b + 2), batch_cb(b + 2));
                compute(qb);
            }
        }
    } else {
        for (int b = 0; b < NB; ++b) {
            compute(qa);
            if (b + 1 < NB) load_batch(qa, batch_g(b + 1), batch_cb(b + 1));
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

// arg-max phase 2 (fused_ops.cu:322) + device-side advance of the autoregressive state:
// the next graph replay consumes st->token / st->pos without a host round trip.
__global__ __launch_bounds__(256) void argmax_final_kernel(const float* __restrict__ pmax,
                                                           const int* __restrict__ pidx, int n,
                                                           StepState* st, uint32_t* ring, int ring_mask,
                                                           int advance) {
    __shared__ float sm[256];
    __shared__ int si[256];
    float b = -INFINITY; int bi = 0x7FFFFFFF;
    for (int i = threadIdx.x; i < n; i += 256) {
        float v = pmax[i]; int ix = pidx[i];
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
        uint32_t t = (uint32_t)si[0];
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
// paged split-KV decode attention (GQA-grouped), D = 128
//   prologue : per-head RMSNorm(q), RMSNorm(k) BEFORE RoPE (modeling.rs:341-359), rotate-half
//              RoPE at `pos` (rotary.rs:372-409), scale q by 1/sqrt(D) (modeling.rs:465),
//              append k,v (bf16 = model dtype) to the paged cache (kv_cache.rs:38-101)
//   main     : online softmax over this block's token range (the math of candle's cpu
//              flash_attn used at modeling.rs:380-420; GQA by integer division)
//   layout   : K/V page = [Hkv][PAGE][D] bf16; a 16-lane row reads one token row
//              (16 lanes x 16 B = 256 B contiguous), 4 rows per wave, 4 waves per block.
// grid (nsplit, Hkv); partial (m, l, o) per (head, split) -> attn_decode_combine_kernel.
// =====================================================================================
template <int NREP, bool KVF32>
__global__ __launch_bounds__(256) void attn_decode_split_kernel(AttnDecArgs a) {
    constexpr int D = 128;
    __shared__ __attribute__((aligned(16))) float qs[NREP][D];
    __shared__ __attribute__((aligned(16))) float knew[D];
    __shared__ __attribute__((aligned(16))) float vnew[D];
    __shared__ float red_m[16][NREP];
    __shared__ float red_l[16][NREP];
    __shared__ __attribute__((aligned(16))) float red_o[16][NREP][D];

    const int split = blockIdx.x, kvh = blockIdx.y, nsplit = gridDim.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane >> 4, sub = lane & 15, dimbase = sub * 8;
    const int Hq = a.Hkv * NREP;

    // Token -> block mapping is INTERLEAVED: chunk j of split s covers tokens
    // [16*(s + nsplit*j), +16), token = chunk base + 4*wave + row.  The first two chunks of
    // every block are therefore known without reading `pos`, so their block-table entries and
    // K/V rows are requested before anything else (pos, q, RoPE tables load meanwhile);
    // validity (t <= pos) is applied as a mask afterwards.  Perfectly balanced for any L.
    const int tok_in_chunk = wave * 4 + r;
    auto kv_off = [&](int t) -> size_t {
        int pi = t / a.page;
        pi = pi < a.max_pages ? pi : a.max_pages - 1;     // speculative loads stay inside the table
        const int page = a.block_table[pi];
        return ((size_t)(page * a.Hkv + kvh) * a.page + (t % a.page)) * D + dimbase;
    };
    // KV element type: bf16 (model dtype, default) or f32 (cm_opts.kv_dtype = CM_KV_F32)
    struct KV8 { u32x4 a, b; };
    auto ld_kv = [&](const void* pool, size_t off) -> KV8 {
        KV8 r;
        if (KVF32) { r.a = ld16((const float*)pool + off); r.b = ld16((const float*)pool + off + 4); }
        else { r.a = ld16((const uint16_t*)pool + off); r.b = r.a; }
        return r;
    };
    KV8 kq[2], vq[2];
    int tt[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        tt[u] = 16 * (split + nsplit * u) + tok_in_chunk;
        const size_t off = kv_off(tt[u]);
        kq[u] = ld_kv(a.kpool, off);
        vq[u] = ld_kv(a.vpool, off);
    }
    const int pos = a.st->pos;
    const int L = pos + 1;
    const bool owner = ((pos >> 4) % nsplit) == split;

    // ---- prologue: q heads of this group, new k, new v ----
    for (int item = wave; item < NREP + 2; item += 4) {
        const float* src;
        const uint16_t* nw = nullptr;
        if (item < NREP) { src = a.qkv + (size_t)(kvh * NREP + item) * D; nw = a.qnw; }
        else if (item == NREP) { src = a.qkv + (size_t)(Hq + kvh) * D; nw = a.knw; }
        else { src = a.qkv + (size_t)(Hq + a.Hkv + kvh) * D; }
        float x1 = src[lane], x2 = src[lane + 64];
        if (item <= NREP) {
            const float c = a.cos[(size_t)pos * (D / 2) + lane];
            const float s = a.sin[(size_t)pos * (D / 2) + lane];
            if (nw != nullptr) {
                const float w1 = bf16_to_f32(nw[lane]), w2 = bf16_to_f32(nw[lane + 64]);
                float ss = wave_sum(x1 * x1 + x2 * x2);
                float rr = 1.0f / sqrtf(ss / (float)D + a.eps);
                x1 = x1 * rr * w1;
                x2 = x2 * rr * w2;
            }
            const float o1 = x1 * c - x2 * s;
            const float o2 = x1 * s + x2 * c;
            x1 = o1; x2 = o2;
        }
        if (item < NREP) {
            qs[item][lane] = x1 * a.scale;
            qs[item][lane + 64] = x2 * a.scale;
        } else {
            float* dst = (item == NREP) ? knew : vnew;
            void* pool = (item == NREP) ? a.kpool : a.vpool;
            const size_t eoff = owner ? ((size_t)(a.block_table[pos / a.page] * a.Hkv + kvh) * a.page + (pos % a.page)) * D : 0;
            if (KVF32) {
                dst[lane] = x1; dst[lane + 64] = x2;
                if (owner) { float* p = (float*)pool + eoff; p[lane] = x1; p[lane + 64] = x2; }
            } else {
                const uint16_t b1 = f32_to_bf16(x1), b2 = f32_to_bf16(x2);
                dst[lane] = bf16_to_f32(b1);
                dst[lane + 64] = bf16_to_f32(b2);
                if (owner) { uint16_t* p = (uint16_t*)pool + eoff; p[lane] = b1; p[lane + 64] = b2; }
            }
        }
    }
    __syncthreads();

    float qr[NREP][8];
#pragma unroll
    for (int h = 0; h < NREP; ++h) {
        const f32x4 q0 = *(const f32x4*)&qs[h][dimbase];
        const f32x4 q1 = *(const f32x4*)&qs[h][dimbase + 4];
        qr[h][0] = q0[0]; qr[h][1] = q0[1]; qr[h][2] = q0[2]; qr[h][3] = q0[3];
        qr[h][4] = q1[0]; qr[h][5] = q1[1]; qr[h][6] = q1[2]; qr[h][7] = q1[3];
    }
    float m[NREP], l[NREP], acc[NREP][8];
#pragma unroll
    for (int h = 0; h < NREP; ++h) {
        m[h] = -INFINITY; l[h] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[h][e] = 0.f;
    }

    auto consume = [&](const KV8& kqv, const KV8& vqv, int t) {
        const bool valid = t < L;
        float kf[8], vf[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (KVF32) {
                kf[e] = __uint_as_float(kqv.a[e]); kf[4 + e] = __uint_as_float(kqv.b[e]);
                vf[e] = __uint_as_float(vqv.a[e]); vf[4 + e] = __uint_as_float(vqv.b[e]);
            } else {
                kf[2 * e] = bf16_lo(kqv.a[e]); kf[2 * e + 1] = bf16_hi(kqv.a[e]);
                vf[2 * e] = bf16_lo(vqv.a[e]); vf[2 * e + 1] = bf16_hi(vqv.a[e]);
            }
        }
        if (t == pos) {   // the token appended by this very step: values from LDS
#pragma unroll
            for (int e = 0; e < 8; ++e) { kf[e] = knew[dimbase + e]; vf[e] = vnew[dimbase + e]; }
        }
#pragma unroll
        for (int h = 0; h < NREP; ++h) {
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) s += qr[h][e] * kf[e];
            s = row16_sum(s);
            if (valid) {
                const float mn = fmaxf(m[h], s);
                const float alpha = expf(m[h] - mn);
                const float p = expf(s - mn);
                l[h] = l[h] * alpha + p;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[h][e] = acc[h][e] * alpha + p * vf[e];
                m[h] = mn;
            }
        }
    };
    consume(kq[0], vq[0], tt[0]);
    consume(kq[1], vq[1], tt[1]);
    // remaining chunks (long contexts): two chunks per iteration, block-uniform bounds
    for (int j = 2; 16 * (split + nsplit * j) < L; j += 2) {
        const bool second = 16 * (split + nsplit * (j + 1)) < L;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            tt[u] = 16 * (split + nsplit * (j + u)) + tok_in_chunk;
            if (u == 0 || second) {
                const size_t off = kv_off(tt[u]);
                kq[u] = ld_kv(a.kpool, off);
                vq[u] = ld_kv(a.vpool, off);
            }
        }
        consume(kq[0], vq[0], tt[0]);
        if (second) consume(kq[1], vq[1], tt[1]);
    }

    // ---- combine the 16 (wave,row) streams of this block ----
    const int slot = wave * 4 + r;
#pragma unroll
    for (int h = 0; h < NREP; ++h) {
        if (sub == 0) { red_m[slot][h] = m[h]; red_l[slot][h] = l[h]; }
        *(f32x4*)&red_o[slot][h][dimbase] = (f32x4){acc[h][0], acc[h][1], acc[h][2], acc[h][3]};
        *(f32x4*)&red_o[slot][h][dimbase + 4] = (f32x4){acc[h][4], acc[h][5], acc[h][6], acc[h][7]};
    }
    __syncthreads();
    for (int it = tid; it < NREP * D; it += 256) {
        const int h = it / D, d = it % D;
        float M = -INFINITY;
#pragma unroll
        for (int i = 0; i < 16; ++i) M = fmaxf(M, red_m[i][h]);
        float O = 0.f, Ls = 0.f;
        if (M > -INFINITY) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float w = expf(red_m[i][h] - M);
                O += w * red_o[i][h][d];
                Ls += w * red_l[i][h];
            }
        }
        const size_t ph = (size_t)(kvh * NREP + h) * nsplit + split;
        a.part_o[ph * D + d] = O;
        if (d == 0) { a.part_ml[ph * 2] = M; a.part_ml[ph * 2 + 1] = Ls; }
    }
}

// grid = Hq, block = 256: out[h, d] = sum_s e^{m_s-M} o_s[d] / sum_s e^{m_s-M} l_s   (nsplit <= 64)
__global__ __launch_bounds__(256) void attn_decode_combine_kernel(const float* __restrict__ part_o,
                                                                  const float* __restrict__ part_ml,
                                                                  float* __restrict__ out, int nsplit) {
    constexpr int D = 128;
    __shared__ float w_s[64];
    __shared__ float inv_l;
    __shared__ float half_o[D];
    const int h = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    if (tid < 64) {
        float mm = -INFINITY, ll = 0.f;
        if (lane < nsplit) {
            const u32x2 v = *(const u32x2*)(part_ml + ((size_t)h * nsplit + lane) * 2);
            mm = __uint_as_float(v[0]); ll = __uint_as_float(v[1]);
        }
        const float M = wave_max(mm);
        const float w = (mm > -INFINITY) ? expf(mm - M) : 0.f;
        const float Ls = wave_sum(w * ll);
        w_s[lane] = w;
        if (lane == 0) inv_l = 1.0f / Ls;
    }
    __syncthreads();
    const int d = tid & (D - 1), half = tid >> 7;
    const int s0 = half * ((nsplit + 1) / 2), s1 = half ? nsplit : (nsplit + 1) / 2;
    float O = 0.f;
    const float* po = part_o + (size_t)h * nsplit * D + d;
#pragma unroll 8
    for (int s = s0; s < s1; ++s) O += w_s[s] * po[(size_t)s * D];
    if (half) half_o[d] = O;
    __syncthreads();
    if (!half) out[(size_t)h * D + d] = (O + half_o[d]) * inv_l;
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
    if (nch % 2 == 0) return {8, 2, 0};
    return {8, 1, 0};
}

template <int PRO, int EPI>
static void launch_gemv_t(const GemvArgs& a, int grid, hipStream_t s) {
    const int nch = (a.K + 511) / 512;
    const size_t lds = (size_t)nch * 512 * 4 + 64;
    const dim3 g(grid), b(256);
    const GemvCfg c = gemv_cfg(a.K);
#define CM_GEMV_CASE(RR, UU, PP) \
    if (c.R == RR && c.U == UU && c.P == PP) { hipLaunchKernelGGL((gemv_bf16_kernel<PRO, EPI, RR, UU, PP != 0, false>), g, b, lds, s, a); return; }
    if (a.K % 512 != 0) { hipLaunchKernelGGL((gemv_bf16_kernel<PRO, EPI, 2, 1, false, true>), g, b, lds, s, a); return; }
    CM_GEMV_CASE(2, 8, 0) CM_GEMV_CASE(2, 8, 1) CM_GEMV_CASE(2, 4, 0) CM_GEMV_CASE(2, 4, 1)
    CM_GEMV_CASE(4, 4, 0) CM_GEMV_CASE(4, 4, 1) CM_GEMV_CASE(4, 2, 0) CM_GEMV_CASE(4, 2, 1)
    CM_GEMV_CASE(8, 2, 0) CM_GEMV_CASE(8, 2, 1) CM_GEMV_CASE(8, 1, 0) CM_GEMV_CASE(8, 1, 1)
    CM_GEMV_CASE(2, 2, 1)
#undef CM_GEMV_CASE
    hipLaunchKernelGGL((gemv_bf16_kernel<PRO, EPI, 8, 1, false, false>), g, b, lds, s, a);
}

int gemv_rows_per_group(int K) { return gemv_cfg(K).R; }

int gemv_grid(int N, int K, int num_cu) {
    // one wave per row group until the chip is full, then grid-stride
    const GemvCfg c = gemv_cfg(K);
    const int groups = (N + c.R - 1) / c.R;
    int blocks = (groups + 3) / 4;
    static int per_cu = -1;
    if (per_cu < 0) { per_cu = 0; if (const char* e = getenv("CM_GEMV_BLOCKS_PER_CU")) per_cu = atoi(e); }
    const int cap = num_cu * (per_cu > 0 ? per_cu : (c.P ? 3 : 4));
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return blocks;
}

void launch_gemv(int pro, int epi, const GemvArgs& a, int grid, hipStream_t s) {
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

void launch_embed_row(const uint16_t* emb, const StepState* st, float* x, int H, int V, hipStream_t s) {
    hipLaunchKernelGGL(embed_row_kernel, dim3((H / 4 + 255) / 256), dim3(256), 0, s, emb, st, x, H, V);
}
void launch_set_state(StepState* st, uint32_t token, int32_t pos, hipStream_t s) {
    hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(1), 0, s, st, token, pos);
}
void launch_argmax_final(const float* pmax, const int* pidx, int n, StepState* st, uint32_t* ring,
                         int ring_mask, int advance, hipStream_t s) {
    hipLaunchKernelGGL(argmax_final_kernel, dim3(1), dim3(256), 0, s, pmax, pidx, n, st, ring, ring_mask, advance);
}

bool launch_attn_decode(const AttnDecArgs& a, int nrep, int nsplit, bool kv_f32, float* out, hipStream_t s) {
    dim3 grid(nsplit, a.Hkv), block(256);
#define CM_ATTN_CASE(N) \
    case N: if (kv_f32) hipLaunchKernelGGL((attn_decode_split_kernel<N, true>), grid, block, 0, s, a); \
            else hipLaunchKernelGGL((attn_decode_split_kernel<N, false>), grid, block, 0, s, a); break;
    switch (nrep) {
        CM_ATTN_CASE(1) CM_ATTN_CASE(2) CM_ATTN_CASE(3) CM_ATTN_CASE(4) CM_ATTN_CASE(6) CM_ATTN_CASE(8)
        default: return false;
    }
#undef CM_ATTN_CASE
    hipLaunchKernelGGL(attn_decode_combine_kernel, dim3(a.Hkv * nrep), dim3(256), 0, s,
                       a.part_o, a.part_ml, out, nsplit);
    return true;
}

}  // namespace cm
