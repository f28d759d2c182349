// Paged split-KV decode attention (GQA-grouped) for head_dim 128 (Qwen3) and 256 (Qwen3.5/3.8).
//
//   prologue : per-head RMSNorm(q), RMSNorm(k) BEFORE RoPE (qwen3/modeling.rs:341-359;
//              qwen3_5/modeling.rs:464-468 with the (1+w) weights folded at load), rotate-half RoPE
//              over the first `rot` dims only (rot == D for Qwen3; partial MRoPE, rot = D/4, for
//              Qwen3.5: qwen3_5/modeling.rs:263-279), q scaled by 1/sqrt(D), K/V appended to the
//              paged cache in the model dtype (modules/kv_cache.rs:38-101, qwen3_5/kv_cache.rs:92-97)
//   main     : online softmax over this block's tokens (the math of candle's cpu flash_attn used at
//              qwen3/modeling.rs:380-420; GQA by integer division)
//   combine  : merge the nsplit partials; Qwen3.5 multiplies by sigmoid(gate) here
//              (qwen3_5/modeling.rs:516-522)
//   layout   : K/V page = [Hkv][PAGE][D]; a token row (D*2 bytes) is read by D/8 lanes x 16 B,
//              64/(D/8) rows per wave, 4 waves per block; grid (nsplit, Hkv).
#include <cstdlib>

#include "dev_common.h"
#include "kernels.h"

namespace cm {

template <int D, int NREP, int KVT>
__global__ __launch_bounds__(256) void attn_decode_split_kernel(AttnDecArgs a) {
    constexpr int LPR = D / 8;            // lanes per token row
    constexpr int RPW = 64 / LPR;         // rows per wave
    constexpr int CT = 4 * RPW;           // tokens per block chunk
    constexpr int EPL = D / 64;           // prologue elements per lane
    __shared__ __attribute__((aligned(16))) float qs[NREP][D];
    __shared__ __attribute__((aligned(16))) float knew[D];
    __shared__ __attribute__((aligned(16))) float vnew[D];
    __shared__ __attribute__((aligned(16))) float tmp[4][D];        // per-wave scratch for the rope pairing
    __shared__ float red_m[CT][NREP];
    __shared__ float red_l[CT][NREP];
    __shared__ __attribute__((aligned(16))) float red_o[CT][NREP][D];

    const int split = blockIdx.x, kvh = blockIdx.y, nsplit = gridDim.x;
    const int bq = blockIdx.z;                    // sequence of a batched step (0 on the single-sequence path)
    const StepState* st = a.st + bq;
    const int32_t* block_table = a.block_table + (size_t)bq * a.bt_stride;
    const float* qkv = a.qkv + (size_t)bq * a.qkv_stride;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane / LPR, sub = lane % LPR, dimbase = sub * 8;

    // Token -> block mapping is INTERLEAVED: chunk j of split s covers tokens
    // [CT*(s + nsplit*j), +CT).  The first two chunks of every block are known without reading
    // `pos`: their block-table entries and K/V rows are requested first (pos, q, RoPE rows load
    // meanwhile); validity (t <= pos) is a mask.  Perfectly balanced for any context length.
    const int tok_in_chunk = wave * RPW + r;
    // KVT: 0 bf16, 4 f16, 1 f32 (element offsets into [pages][Hkv][PAGE][D]); 2 int8, 3 int4 (per-token symmetric codes +
    // f32 scale, qwen3_5/kv_cache.rs:253-301; byte offsets, each page = codes [Hkv][PAGE][row] then scales [Hkv][PAGE])
    constexpr bool KVQ = KVT == KV_INT8 || KVT == KV_INT4;
    constexpr bool KVF32 = KVT == 1;
    constexpr int ROWB = KVT == 2 ? D : D / 2;            // code bytes per token row (quantised modes)
    auto kv_off = [&](int t) -> size_t {
        int pi = t / a.page;
        pi = pi < a.max_pages ? pi : a.max_pages - 1;     // speculative loads stay inside the table
        const int page = block_table[pi];
        if (KVQ) return (size_t)page * a.page_bytes + (size_t)(kvh * a.page + (t % a.page)) * ROWB;
        return ((size_t)(page * a.Hkv + kvh) * a.page + (t % a.page)) * D + dimbase;
    };
    auto sc_off = [&](int t) -> size_t {                  // byte offset of the row's f32 scale
        int pi = t / a.page;
        pi = pi < a.max_pages ? pi : a.max_pages - 1;
        return (size_t)block_table[pi] * a.page_bytes + (size_t)a.Hkv * a.page * ROWB + (size_t)(kvh * a.page + (t % a.page)) * 4;
    };
    struct KV8 { u32x4 a, b; float s; };
    auto ld_kv = [&](const void* pool, size_t off, int t) -> KV8 {
        KV8 v;
        v.s = 1.f;
        if (KVT == 1) { v.a = ld16((const float*)pool + off); v.b = ld16((const float*)pool + off + 4); }
        else if (!KVQ) { v.a = ld16((const uint16_t*)pool + off); v.b = v.a; }
        else {
            const uint8_t* p = (const uint8_t*)pool;
            if (KVT == 2) { const u32x2 c = *(const u32x2*)(p + off + dimbase); v.a = (u32x4){c[0], c[1], 0, 0}; }
            else v.a = (u32x4){*(const uint32_t*)(p + off + (dimbase >> 1)), 0, 0, 0};
            v.b = v.a;
            v.s = *(const float*)(p + sc_off(t));
        }
        return v;
    };
    KV8 kq[2], vq[2];
    int tt[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        tt[u] = CT * (split + nsplit * u) + tok_in_chunk;
        const size_t off = kv_off(tt[u]);
        kq[u] = ld_kv(a.kpool, off, tt[u]);
        vq[u] = ld_kv(a.vpool, off, tt[u]);
    }
    const int pos = st->pos;
    const int rpos = pos + st->rsv[0];          // rotary position = cache position + MRoPE delta (vlm.rs:294-301)
    const int L = pos + 1;
    const bool owner = ((pos / CT) % nsplit) == split;
    const int rot = a.rot_dim, hrot = rot >> 1;

    // ---- prologue: q heads of this group, new k, new v (one wave per item) ----
    for (int item = wave; item < NREP + 2; item += 4) {
        const float* src;
        const float* nw = nullptr;
        if (item < NREP) { src = qkv + a.q_off + (size_t)(kvh * NREP + item) * D; nw = a.qnw; }
        else if (item == NREP) { src = qkv + a.k_off + (size_t)kvh * D; nw = a.knw; }
        else { src = qkv + a.v_off + (size_t)kvh * D; }
        float xv[EPL];
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < EPL; ++j) { xv[j] = src[lane + 64 * j]; ss += xv[j] * xv[j]; }
        if (item <= NREP) {
            if (nw != nullptr) {
                ss = wave_sum(ss);
                const float rr = 1.0f / sqrtf(ss / (float)D + a.eps);
#pragma unroll
                for (int j = 0; j < EPL; ++j) xv[j] = xv[j] * rr * nw[lane + 64 * j];
            }
            // rotate-half inside the first `rot` dims: partner of d is d +/- rot/2
#pragma unroll
            for (int j = 0; j < EPL; ++j) tmp[wave][lane + 64 * j] = xv[j];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                const int d = lane + 64 * j;
                if (d < rot) {
                    const int i = d < hrot ? d : d - hrot;
                    const float c = a.cos[(size_t)rpos * hrot + i], s = a.sin[(size_t)rpos * hrot + i];
                    const float lo = tmp[wave][i], hi = tmp[wave][i + hrot];
                    xv[j] = d < hrot ? lo * c - hi * s : lo * s + hi * c;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (item < NREP) {
#pragma unroll
            for (int j = 0; j < EPL; ++j) qs[item][lane + 64 * j] = xv[j] * a.scale;
        } else {
            float* dst = (item == NREP) ? knew : vnew;
            void* pool = (item == NREP) ? a.kpool : a.vpool;
            if (KVQ) {
                // quantize_per_token (kv_cache.rs:253-268): scale = amax * (1/qmax) + 1e-8, code = round(x / scale) + offset;
                // attention reads the DEQUANTISED value of the new token too (append() returns dequantize(full cache), :318-322)
                constexpr float QMAX = KVT == 2 ? 127.f : 7.f, OFFS = KVT == 2 ? 128.f : 8.f;
                float amax = 0.f;
#pragma unroll
                for (int j = 0; j < EPL; ++j) amax = fmaxf(amax, fabsf(xv[j]));
                amax = wave_max(amax);
                const float scale = __fadd_rn(__fmul_rn(amax, (float)(1.0 / (double)QMAX)), 1e-8f);
                uint8_t* pb = (uint8_t*)pool;
                const size_t roff = owner ? kv_off(pos) : 0;
#pragma unroll
                for (int j = 0; j < EPL; ++j) {
                    const int d = lane + 64 * j;
                    const float code = roundf(__fdiv_rn(xv[j], scale)) + OFFS;
                    dst[d] = __fmul_rn(code - OFFS, scale);
                    const uint32_t ci = (uint32_t)(int)code;
                    if (KVT == 2) { if (owner) pb[roff + d] = (uint8_t)ci; }
                    else {
                        const uint32_t hi = (uint32_t)__shfl_down((int)ci, 1);      // byte = lo + hi*16 for (even, odd) pairs
                        if (owner && !(lane & 1)) pb[roff + (d >> 1)] = (uint8_t)(ci | (hi << 4));
                    }
                }
                if (owner && lane == 0) *(float*)(pb + sc_off(pos)) = scale;
            } else {
            const size_t eoff = owner ? ((size_t)(block_table[pos / a.page] * a.Hkv + kvh) * a.page + (pos % a.page)) * D : 0;
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                const int d = lane + 64 * j;
                if (KVF32) {
                    dst[d] = xv[j];
                    if (owner) ((float*)pool)[eoff + d] = xv[j];
                } else {
                    const uint16_t b = kv16_from_f32<KVT>(xv[j]);
                    dst[d] = kv16_to_f32<KVT>(b);
                    if (owner) ((uint16_t*)pool)[eoff + d] = b;
                }
            }
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
            if (KVT == 2) {           // dequantize_per_token: (code - 128) * scale
                kf[e] = __fmul_rn((float)((kqv.a[0] >> (8 * e)) & 0xFFu) - 128.f, kqv.s);
                kf[4 + e] = __fmul_rn((float)((kqv.a[1] >> (8 * e)) & 0xFFu) - 128.f, kqv.s);
                vf[e] = __fmul_rn((float)((vqv.a[0] >> (8 * e)) & 0xFFu) - 128.f, vqv.s);
                vf[4 + e] = __fmul_rn((float)((vqv.a[1] >> (8 * e)) & 0xFFu) - 128.f, vqv.s);
            } else if (KVT == 3) {    // nibbles: element 2i = low, 2i+1 = high of byte i
                kf[2 * e] = __fmul_rn((float)((kqv.a[0] >> (8 * e)) & 0xFu) - 8.f, kqv.s);
                kf[2 * e + 1] = __fmul_rn((float)((kqv.a[0] >> (8 * e + 4)) & 0xFu) - 8.f, kqv.s);
                vf[2 * e] = __fmul_rn((float)((vqv.a[0] >> (8 * e)) & 0xFu) - 8.f, vqv.s);
                vf[2 * e + 1] = __fmul_rn((float)((vqv.a[0] >> (8 * e + 4)) & 0xFu) - 8.f, vqv.s);
            } else if (KVF32) {
                kf[e] = __uint_as_float(kqv.a[e]); kf[4 + e] = __uint_as_float(kqv.b[e]);
                vf[e] = __uint_as_float(vqv.a[e]); vf[4 + e] = __uint_as_float(vqv.b[e]);
            } else {
                kf[2 * e] = kv16_lo<KVT>(kqv.a[e]); kf[2 * e + 1] = kv16_hi<KVT>(kqv.a[e]);
                vf[2 * e] = kv16_lo<KVT>(vqv.a[e]); vf[2 * e + 1] = kv16_hi<KVT>(vqv.a[e]);
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
            if (LPR == 32) s += __shfl_xor(s, 16);
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
    for (int j = 2; CT * (split + nsplit * j) < L; j += 2) {
        const bool second = CT * (split + nsplit * (j + 1)) < L;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            tt[u] = CT * (split + nsplit * (j + u)) + tok_in_chunk;
            if (u == 0 || second) {
                const size_t off = kv_off(tt[u]);
                kq[u] = ld_kv(a.kpool, off, tt[u]);
                vq[u] = ld_kv(a.vpool, off, tt[u]);
            }
        }
        consume(kq[0], vq[0], tt[0]);
        if (second) consume(kq[1], vq[1], tt[1]);
    }

    // ---- combine the CT (wave,row) streams of this block ----
    const int slot = wave * RPW + r;
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
        for (int i = 0; i < CT; ++i) M = fmaxf(M, red_m[i][h]);
        float O = 0.f, Ls = 0.f;
        if (M > -INFINITY) {
#pragma unroll
            for (int i = 0; i < CT; ++i) {
                const float w = expf(red_m[i][h] - M);
                O += w * red_o[i][h][d];
                Ls += w * red_l[i][h];
            }
        }
        if (a.out1 != nullptr) {                                 // the only split of this sequence: what the combine kernel would compute
            float v = O * (1.0f / Ls);                           //   from one partial (weight e^0 = 1), bit for bit
            if (a.gate != nullptr) v *= 1.0f / (1.0f + expf(-a.gate[(size_t)bq * a.qkv_stride + (size_t)(kvh * NREP + h) * D + d]));
            if (a.out1_hi != nullptr) {
                const size_t off = (size_t)bq * a.out1_cols + (size_t)(kvh * NREP + h) * D + d;
                const uint16_t hh = f32_to_bf16(v);
                a.out1_hi[off] = hh;
                a.out1_lo[off] = f32_to_bf16(v - bf16_to_f32(hh));
            } else {
                a.out1[(size_t)bq * a.out1_stride + (size_t)(kvh * NREP + h) * D + d] = v;
            }
            continue;
        }
        const size_t ph = ((size_t)bq * a.Hkv * NREP + (size_t)(kvh * NREP + h)) * nsplit + split;
        a.part_o[ph * D + d] = O;
        if (d == 0) { a.part_ml[ph * 2] = M; a.part_ml[ph * 2 + 1] = Ls; }
    }
}

// grid = Hq, block = 1024: out[h, d] = sum_s e^{m_s-M} o_s[d] / sum_s e^{m_s-M} l_s  (* sigmoid(gate)).
// The kernel is pure latency (a few KB per head): 1024 / D thread groups split the `s` range so that every partial a
// thread needs is requested in ONE batch, before the (m, l) weights it will be scaled by have even arrived.
template <int D>
__global__ __launch_bounds__(1024) void attn_decode_combine_kernel(const float* __restrict__ part_o,
                                                                   const float* __restrict__ part_ml,
                                                                   const float* __restrict__ gate,
                                                                   float* __restrict__ out, int nsplit, int gate_stride,
                                                                   int out_stride) {
    constexpr int NH = 1024 / D;                 // thread groups that split the `s` range
    constexpr int PER = 64 / NH;                 // nsplit <= 64
    __shared__ float w_s[64];
    __shared__ float inv_l;
    __shared__ float grp_o[NH][D];
    const int h = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int bq = blockIdx.y, nh = gridDim.x;
    part_o += (size_t)bq * nh * nsplit * D;
    part_ml += (size_t)bq * nh * nsplit * 2;
    out += (size_t)bq * out_stride;
    if (gate != nullptr) gate += (size_t)bq * gate_stride;
    const int d = tid % D, grp = tid / D;
    const int per = (nsplit + NH - 1) / NH;
    const int s0 = grp * per, s1 = min(nsplit, s0 + per);
    const float* po = part_o + (size_t)h * nsplit * D + d;
    float pv[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) pv[j] = (s0 + j < s1) ? po[(size_t)(s0 + j) * D] : 0.f;
    float gv = 0.f;
    if (gate != nullptr && grp == 0) gv = gate[(size_t)h * D + d];
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
    float O = 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j)
        if (s0 + j < s1) O += w_s[s0 + j] * pv[j];
    grp_o[grp][d] = O;
    __syncthreads();
    if (grp) return;
#pragma unroll
    for (int g = 1; g < NH; ++g) O += grp_o[g][d];
    float v = O * inv_l;
    if (gate != nullptr) v *= 1.0f / (1.0f + expf(-gv));
    out[(size_t)h * D + d] = v;
}


// ---------------------------------------------------------------------------------------------------------
// Short/medium-context variant: ONE 1024-thread block per (token split, q head).  At context ~1K the split
// kernel above is pure fixed latency (2 chunks per block) followed by a second fixed-latency combine launch;
// here 16 waves keep PF chunks of K/V rows in flight each, the CT row streams are merged inside the block, and
// the few (NS <= 4) per-head partials are merged by the o_proj GEMV's x-staging prologue (PRO_ATTNCOMB) -- the
// combine launch disappears.  blockIdx.y -> (kv head = y % Hkv, head-in-group = y / Hkv): blocks round-robin
// over the 8 XCDs, so with Hkv = 8 the NREP q heads that share a KV head run on the SAME XCD and share its L2.
// ---------------------------------------------------------------------------------------------------------
template <int D, bool KVF32>
__global__ __launch_bounds__(1024) void attn_decode_head_kernel(AttnDecArgs a, int nrep) {
    constexpr int LPR = D / 8, RPW = 64 / LPR, NW = 16, CT = NW * RPW, EPL = D / 64;
    constexpr int PF = KVF32 ? 2 : 4;                 // chunks of K/V rows in flight per wave
    constexpr int NP = 1024 / D;                      // thread groups of the in-block merge
    __shared__ __attribute__((aligned(16))) float qs[D];
    __shared__ __attribute__((aligned(16))) float knew[D];
    __shared__ __attribute__((aligned(16))) float vnew[D];
    __shared__ __attribute__((aligned(16))) float tmp[3][D];
    __shared__ float red_m[CT];
    __shared__ float red_l[CT];
    __shared__ __attribute__((aligned(16))) float red_o[CT][D];
    __shared__ float red2[NP][D];
    __shared__ float red2l[NP];

    const int split = blockIdx.x, NS = gridDim.x;
    const int kvh = blockIdx.y % a.Hkv, hin = blockIdx.y / a.Hkv, head = kvh * nrep + hin;
    const int bq = blockIdx.z;
    const StepState* st = a.st + bq;
    const int32_t* block_table = a.block_table + (size_t)bq * a.bt_stride;
    const float* qkv = a.qkv + (size_t)bq * a.qkv_stride;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane / LPR, sub = lane % LPR, dimbase = sub * 8;
    const int tok_in_chunk = wave * RPW + r;
    auto kv_off = [&](int t) -> size_t {
        int pi = t / a.page;
        pi = pi < a.max_pages ? pi : a.max_pages - 1;
        const int page = block_table[pi];
        return ((size_t)(page * a.Hkv + kvh) * a.page + (t % a.page)) * D + dimbase;
    };
    struct KV8 { u32x4 a, b; };
    auto ld_kv = [&](const void* pool, size_t off) -> KV8 {
        KV8 v;
        if (KVF32) { v.a = ld16((const float*)pool + off); v.b = ld16((const float*)pool + off + 4); }
        else { v.a = ld16((const uint16_t*)pool + off); v.b = v.a; }
        return v;
    };
    KV8 kq[PF], vq[PF];
    int tt[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {                     // speculative: independent of pos
        tt[u] = CT * (split + NS * u) + tok_in_chunk;
        const size_t off = kv_off(tt[u]);
        kq[u] = ld_kv(a.kpool, off);
        vq[u] = ld_kv(a.vpool, off);
    }
    const int pos = st->pos;
    const int rpos = pos + st->rsv[0];
    const int L = pos + 1;
    const bool owner = hin == 0 && ((pos / CT) % NS) == split;
    const int rot = a.rot_dim, hrot = rot >> 1;

    if (wave < 3) {                                    // wave 0: q head, wave 1: new k, wave 2: new v
        const int item = wave;
        const float* src;
        const float* nw = nullptr;
        if (item == 0) { src = qkv + a.q_off + (size_t)head * D; nw = a.qnw; }
        else if (item == 1) { src = qkv + a.k_off + (size_t)kvh * D; nw = a.knw; }
        else { src = qkv + a.v_off + (size_t)kvh * D; }
        float xv[EPL];
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < EPL; ++j) { xv[j] = src[lane + 64 * j]; ss += xv[j] * xv[j]; }
        if (item <= 1) {
            if (nw != nullptr) {
                ss = wave_sum(ss);
                const float rr = 1.0f / sqrtf(ss / (float)D + a.eps);
#pragma unroll
                for (int j = 0; j < EPL; ++j) xv[j] = xv[j] * rr * nw[lane + 64 * j];
            }
#pragma unroll
            for (int j = 0; j < EPL; ++j) tmp[item][lane + 64 * j] = xv[j];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                const int d = lane + 64 * j;
                if (d < rot) {
                    const int i = d < hrot ? d : d - hrot;
                    const float c = a.cos[(size_t)rpos * hrot + i], s = a.sin[(size_t)rpos * hrot + i];
                    const float lo = tmp[item][i], hi = tmp[item][i + hrot];
                    xv[j] = d < hrot ? lo * c - hi * s : lo * s + hi * c;
                }
            }
        }
        if (item == 0) {
#pragma unroll
            for (int j = 0; j < EPL; ++j) qs[lane + 64 * j] = xv[j] * a.scale;
        } else {
            float* dst = (item == 1) ? knew : vnew;
            void* pool = (item == 1) ? a.kpool : a.vpool;
            const size_t eoff = owner ? ((size_t)(block_table[pos / a.page] * a.Hkv + kvh) * a.page + (pos % a.page)) * D : 0;
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                const int d = lane + 64 * j;
                if (KVF32) {
                    dst[d] = xv[j];
                    if (owner) ((float*)pool)[eoff + d] = xv[j];
                } else {
                    const uint16_t b = f32_to_bf16(xv[j]);
                    dst[d] = bf16_to_f32(b);
                    if (owner) ((uint16_t*)pool)[eoff + d] = b;
                }
            }
        }
    }
    __syncthreads();

    float qr[8];
    {
        const f32x4 q0 = *(const f32x4*)&qs[dimbase];
        const f32x4 q1 = *(const f32x4*)&qs[dimbase + 4];
        qr[0] = q0[0]; qr[1] = q0[1]; qr[2] = q0[2]; qr[3] = q0[3];
        qr[4] = q1[0]; qr[5] = q1[1]; qr[6] = q1[2]; qr[7] = q1[3];
    }
    float m = -INFINITY, l = 0.f, acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;

    auto consume = [&](const KV8& kqv, const KV8& vqv, int t) {
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
        if (t == pos) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { kf[e] = knew[dimbase + e]; vf[e] = vnew[dimbase + e]; }
        }
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += qr[e] * kf[e];
        s = row16_sum(s);
        if (LPR == 32) s += __shfl_xor(s, 16);
        if (t < L) {
            const float mn = fmaxf(m, s);
            const float alpha = expf(m - mn);
            const float p = expf(s - mn);
            l = l * alpha + p;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = acc[e] * alpha + p * vf[e];
            m = mn;
        }
    };
#pragma unroll
    for (int u = 0; u < PF; ++u) consume(kq[u], vq[u], tt[u]);
    for (int j = PF; CT * (split + NS * j) < L; j += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            tt[u] = CT * (split + NS * (j + u)) + tok_in_chunk;
            if (CT * (split + NS * (j + u)) < L) {            // block-uniform
                const size_t off = kv_off(tt[u]);
                kq[u] = ld_kv(a.kpool, off);
                vq[u] = ld_kv(a.vpool, off);
            }
        }
#pragma unroll
        for (int u = 0; u < PF; ++u)
            if (CT * (split + NS * (j + u)) < L) consume(kq[u], vq[u], tt[u]);
    }

    // ---- merge the CT row streams of this block ----
    const int slot = wave * RPW + r;
    if (sub == 0) { red_m[slot] = m; red_l[slot] = l; }
    *(f32x4*)&red_o[slot][dimbase] = (f32x4){acc[0], acc[1], acc[2], acc[3]};
    *(f32x4*)&red_o[slot][dimbase + 4] = (f32x4){acc[4], acc[5], acc[6], acc[7]};
    __syncthreads();
    const int d = tid % D, part = tid / D;
    float M = -INFINITY;
#pragma unroll 8
    for (int i = 0; i < CT; ++i) M = fmaxf(M, red_m[i]);
    float O = 0.f, Ls = 0.f;
    if (M > -INFINITY) {
        constexpr int PER = CT / NP;
#pragma unroll
        for (int i = part * PER; i < (part + 1) * PER; ++i) {
            const float w = expf(red_m[i] - M);
            O += w * red_o[i][d];
            Ls += w * red_l[i];
        }
    }
    red2[part][d] = O;
    if (d == 0) red2l[part] = Ls;
    __syncthreads();
    if (part == 0) {
        float Ot = 0.f, Lt = 0.f;
#pragma unroll
        for (int p = 0; p < NP; ++p) { Ot += red2[p][d]; Lt += red2l[p]; }
        const size_t ph = ((size_t)bq * a.Hkv * nrep + (size_t)head) * NS + split;
        a.part_o[ph * D + d] = Ot;
        if (d == 0) { a.part_ml[ph * 2] = M; a.part_ml[ph * 2 + 1] = Lt; }
    }
}

bool launch_attn_decode_heads(const AttnDecArgs& a, int D, int nrep, int ns, bool kv_f32, int n_seq, hipStream_t s) {
    dim3 grid(ns, a.Hkv * nrep, n_seq), block(1024);
    if (D == 128) {
        if (kv_f32) hipLaunchKernelGGL((attn_decode_head_kernel<128, true>), grid, block, 0, s, a, nrep);
        else hipLaunchKernelGGL((attn_decode_head_kernel<128, false>), grid, block, 0, s, a, nrep);
    } else if (D == 256) {
        if (kv_f32) hipLaunchKernelGGL((attn_decode_head_kernel<256, true>), grid, block, 0, s, a, nrep);
        else hipLaunchKernelGGL((attn_decode_head_kernel<256, false>), grid, block, 0, s, a, nrep);
    } else {
        return false;
    }
    return true;
}


// ---------------------------------------------------------------------------------------------------------
// Long-context variant on the matrix cores (bf16 KV).  The split kernel above spends ~35 VALU ops per (token, lane):
// a 16-lane dot-product reduction per head and an exp replicated over the 16 lanes of a row -- at 32 K tokens it is
// VALU-bound at 1.8 TB/s.  Here the NREP query heads that share a KV head are the 16 "query columns" of an MFMA tile
// (the prefill kernel's transposed formulation):
//   S^T[16 tokens x 16 heads] = K_tile . Q^T     mfma_f32_16x16x32_bf16, A = K rows straight from the pages (16 B per
//                                                lane), B = q as bf16 hi + lo (two MFMAs: no q rounding loss)
//   online softmax per head column (statistics replicated over the 4 row groups of a lane's column)
//   O^T[D x 16 heads] += V^T . P^T               mfma_f32_16x16x16bf16_1k, A = V^T through ds_read_tr16_b64 from the wave's
//                                                private LDS tile, B = P^T as bf16 hi + lo straight from the S^T registers
// One wave owns 16 tokens per step (a block covers 64), the next step's K and V rows are prefetched into a second
// register set; grid (nsplit, Hkv); partials go to the same combine kernel.
// ---------------------------------------------------------------------------------------------------------
// KVT = 2 / 3 (int8 / int4 pages): the codes minus their offset are small integers -- exact in bf16 -- so they feed the
// MFMAs directly; the per-token scales are applied outside the products: S^T rows are multiplied by the K scale of
// their token and p by the V scale before it becomes the P^T operand (l uses the unscaled p).
template <int D, int NREP, int KVT>
__global__ __launch_bounds__(256) void attn_decode_mfma_kernel(AttnDecArgs a) {
    constexpr bool KVQ = KVT == KV_INT8 || KVT == KV_INT4;
    constexpr int ROWB = KVT == 2 ? D : D / 2;                   // code bytes per token row
    constexpr float OFFS = KVT == 2 ? 128.f : 8.f, QMAX = KVT == 2 ? 127.f : 7.f;
    constexpr int NKS = D / 32, NNT = D / 16, VLD = D + 16, EPL = D / 64, TB = 64;
    constexpr int VCH = (16 * D / 8) / 64;                       // 16-byte V chunks per lane per tile
    __shared__ __attribute__((aligned(16))) uint16_t q_hi[16 * D];
    __shared__ __attribute__((aligned(16))) uint16_t q_lo[16 * D];
    __shared__ __attribute__((aligned(16))) uint16_t knew[D];
    __shared__ __attribute__((aligned(16))) uint16_t vnew[D];
    __shared__ __attribute__((aligned(16))) float tmp[4][D];
    __shared__ __attribute__((aligned(16))) uint16_t Vs[4][16 * VLD];      // one tile per wave; reused as red_o at the end
    __shared__ float red_m[4][16];
    __shared__ float red_l[4][16];
    __shared__ float new_scale[2];                               // quantised KV: scales of the appended k / v rows

    const int split = blockIdx.x, kvh = blockIdx.y, nsplit = gridDim.x, bq = blockIdx.z;
    const StepState* st = a.st + bq;
    const int32_t* block_table = a.block_table + (size_t)bq * a.bt_stride;
    const float* qkv = a.qkv + (size_t)bq * a.qkv_stride;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = lane & 15, g = lane >> 4;
    const uint16_t* kpool = (const uint16_t*)a.kpool;
    const uint16_t* vpool = (const uint16_t*)a.vpool;

    for (int i = tid; i < 16 * D; i += 256) { q_hi[i] = 0; q_lo[i] = 0; }
    const int pos = st->pos;
    const int rpos = pos + st->rsv[0];
    const bool owner = ((pos / TB) % nsplit) == split;
    const int rot = a.rot_dim, hrot = rot >> 1;
    __syncthreads();

    // this wave's tiles: CACHED tokens [TB * (split + nsplit * j) + 16 * wave, +16), j = 0, 1, ... (t < pos: written by
    // earlier steps, visible); the token appended by this very step is one extra pseudo-tile of the owner block
    const int psh = __builtin_ctz(a.page);                      // page size is a power of two (launcher checks)
    auto row_off = [&](int t) -> size_t {
        const int page = block_table[t >> psh];
        if (KVQ) return (size_t)page * a.page_bytes + (size_t)(kvh * a.page + (t & (a.page - 1))) * ROWB;     // bytes
        return ((size_t)(page * a.Hkv + kvh) * a.page + (t & (a.page - 1))) * D;
    };
    auto scale_off = [&](int t) -> size_t {                      // byte offset of token t's f32 scale
        return (size_t)block_table[t >> psh] * a.page_bytes + (size_t)a.Hkv * a.page * ROWB + (size_t)(kvh * a.page + (t & (a.page - 1))) * 4;
    };
    // 8 codes -> 8 bf16 integers (code - offset)
    auto codes8 = [&](uint32_t lo, uint32_t hi) -> u32x4 {       // int8: lo / hi = bytes 0..3 / 4..7; int4: lo = 8 nibbles
        float f[8];
        if (KVT == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { f[e] = (float)((lo >> (8 * e)) & 0xFFu) - OFFS; f[4 + e] = (float)((hi >> (8 * e)) & 0xFFu) - OFFS; }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = (float)((lo >> (4 * e)) & 0xFu) - OFFS;
        }
        return (u32x4){pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7])};
    };
    f32x4 ksA, ksB, vsA, vsB;                                    // per-token scales of the tile rows g*4 .. g*4+3
    auto load_tile = [&](bf16x8 (&kf)[NKS], u32x4 (&vf)[VCH], f32x4& kscl, f32x4& vscl, int tb) {
        const size_t ko = row_off(min(tb + sub, pos - 1));               // clamped rows are masked later
        if (KVQ) {
            const uint8_t* kb = (const uint8_t*)a.kpool + ko;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const int d0 = ks * 32 + g * 8;
                u32x4 c;
                if (KVT == 2) { const u32x2 w = *(const u32x2*)(kb + d0); c = codes8(w[0], w[1]); }
                else c = codes8(*(const uint32_t*)(kb + (d0 >> 1)), 0);
                kf[ks] = __builtin_bit_cast(bf16x8, c);
            }
#pragma unroll
            for (int i = 0; i < VCH; ++i) {
                const int cch = lane + 64 * i, tok = cch / (D / 8), d8 = (cch % (D / 8)) * 8;
                const uint8_t* vb = (const uint8_t*)a.vpool + row_off(min(tb + tok, pos - 1));
                if (KVT == 2) { const u32x2 w = *(const u32x2*)(vb + d8); vf[i] = codes8(w[0], w[1]); }
                else vf[i] = codes8(*(const uint32_t*)(vb + (d8 >> 1)), 0);
            }
            // rows g*4 .. g*4+3 of a 16-aligned tile sit in one page: 4 contiguous scales (clamped tokens are masked)
            const int t0 = min(tb + g * 4, pos - 1), t3 = min(tb + g * 4 + 3, pos - 1);
            if (t3 - t0 == 3) {
                kscl = *(const f32x4*)((const uint8_t*)a.kpool + scale_off(t0));
                vscl = *(const f32x4*)((const uint8_t*)a.vpool + scale_off(t0));
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int t = min(tb + g * 4 + r, pos - 1);
                    kscl[r] = *(const float*)((const uint8_t*)a.kpool + scale_off(t));
                    vscl[r] = *(const float*)((const uint8_t*)a.vpool + scale_off(t));
                }
            }
            return;
        }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) kf[ks] = *(const bf16x8*)(kpool + ko + g * 8 + ks * 32);
#pragma unroll
        for (int i = 0; i < VCH; ++i) {
            const int c = lane + 64 * i, tok = c / (D / 8), d8 = (c % (D / 8)) * 8;
            vf[i] = ld16(vpool + row_off(min(tb + tok, pos - 1)) + d8);
        }
    };
    // the first tile's K / V rows are requested BEFORE the q prologue: at short contexts a wave has a single tile, and
    // its HBM latency then overlaps the q loads, norms and rotations instead of following them
    bf16x8 kA[NKS], kB[NKS];
    u32x4 vA[VCH], vB[VCH];
    const int tb_first = TB * split + 16 * wave;
    if constexpr (D <= 128 && !KVQ) {
        if (tb_first < pos) load_tile(kA, vA, ksA, vsA, tb_first);
    }

    // ---- prologue: the group's q heads (norm, rope, 1/sqrt(D), bf16 hi + lo), new k / v (bf16; owner appends) ----
    for (int item = wave; item < NREP + 2; item += 4) {
        const float* src;
        const float* nw = nullptr;
        if (item < NREP) { src = qkv + a.q_off + (size_t)(kvh * NREP + item) * D; nw = a.qnw; }
        else if (item == NREP) { src = qkv + a.k_off + (size_t)kvh * D; nw = a.knw; }
        else { src = qkv + a.v_off + (size_t)kvh * D; }
        float xv[EPL];
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < EPL; ++j) xv[j] = src[lane + 64 * j];
        if (a.qkv_ns > 1) {
            // the rows are the K-split partials of the int8 qkv GEMM (2 .. 4 slices): added here in slice order.  All loads first (a
            // missing slice re-reads slice 0 and adds nothing): one round trip, not one per slice
            float p[3][EPL];
#pragma unroll
            for (int sl = 1; sl < 4; ++sl)
#pragma unroll
                for (int j = 0; j < EPL; ++j) p[sl - 1][j] = src[(size_t)(sl < a.qkv_ns ? sl : 0) * a.qkv_slice + lane + 64 * j];
#pragma unroll
            for (int sl = 1; sl < 4; ++sl)
#pragma unroll
                for (int j = 0; j < EPL; ++j) xv[j] += sl < a.qkv_ns ? p[sl - 1][j] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < EPL; ++j) ss += xv[j] * xv[j];
        if (item <= NREP) {
            if (nw != nullptr) {
                ss = wave_sum(ss);
                const float rr = 1.0f / sqrtf(ss / (float)D + a.eps);
#pragma unroll
                for (int j = 0; j < EPL; ++j) xv[j] = xv[j] * rr * nw[lane + 64 * j];
            }
#pragma unroll
            for (int j = 0; j < EPL; ++j) tmp[wave][lane + 64 * j] = xv[j];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                const int d = lane + 64 * j;
                if (d < rot) {
                    const int i = d < hrot ? d : d - hrot;
                    const float c = a.cos[(size_t)rpos * hrot + i], sn = a.sin[(size_t)rpos * hrot + i];
                    const float lo = tmp[wave][i], hi = tmp[wave][i + hrot];
                    xv[j] = d < hrot ? lo * c - hi * sn : lo * sn + hi * c;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (item < NREP) {
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                const float x = xv[j] * a.scale;
                const uint16_t h = kv16_from_f32<KVT>(x);
                q_hi[item * D + lane + 64 * j] = h;
                q_lo[item * D + lane + 64 * j] = kv16_from_f32<KVT>(x - kv16_to_f32<KVT>(h));
            }
        } else {
            uint16_t* dst = (item == NREP) ? knew : vnew;
            if (KVQ) {
                // quantize_per_token (qwen3_5/kv_cache.rs:253-268); LDS keeps code - offset as a bf16 integer
                float amax = 0.f;
#pragma unroll
                for (int j = 0; j < EPL; ++j) amax = fmaxf(amax, fabsf(xv[j]));
                amax = wave_max(amax);
                const float scale = __fadd_rn(__fmul_rn(amax, (float)(1.0 / (double)QMAX)), 1e-8f);
                uint8_t* pb = (uint8_t*)((item == NREP) ? a.kpool : a.vpool);
                const size_t pbase = owner ? (size_t)block_table[pos / a.page] * a.page_bytes : 0;
                const size_t roff = pbase + (size_t)(kvh * a.page + (pos % a.page)) * ROWB;
#pragma unroll
                for (int j = 0; j < EPL; ++j) {
                    const int d = lane + 64 * j;
                    const float code = roundf(__fdiv_rn(xv[j], scale)) + OFFS;
                    dst[d] = f32_to_bf16(code - OFFS);
                    const uint32_t ci = (uint32_t)(int)code;
                    if (KVT == 2) { if (owner) pb[roff + d] = (uint8_t)ci; }
                    else {
                        const uint32_t hi = (uint32_t)__shfl_down((int)ci, 1);
                        if (owner && !(lane & 1)) pb[roff + (d >> 1)] = (uint8_t)(ci | (hi << 4));
                    }
                }
                if (lane == 0) {
                    new_scale[item - NREP] = scale;
                    if (owner) *(float*)(pb + pbase + (size_t)a.Hkv * a.page * ROWB + (size_t)(kvh * a.page + (pos % a.page)) * 4) = scale;
                }
            } else {
            uint16_t* pool = (uint16_t*)((item == NREP) ? a.kpool : a.vpool);
            const size_t eoff = owner ? ((size_t)(block_table[pos / a.page] * a.Hkv + kvh) * a.page + (pos % a.page)) * D : 0;
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                const int d = lane + 64 * j;
                const uint16_t b = kv16_from_f32<KVT>(xv[j]);
                dst[d] = b;
                if (owner) pool[eoff + d] = b;
            }
            }
        }
    }
    __syncthreads();

    // Q^T fragments (B operand): lane holds q[head = sub][dims g*8 + 32*ks .. +8]
    bf16x8 qh[NKS], ql[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        qh[ks] = *(const bf16x8*)&q_hi[sub * D + ks * 32 + g * 8];
        ql[ks] = *(const bf16x8*)&q_lo[sub * D + ks * 32 + g * 8];
    }
    f32x4 o[NNT];
#pragma unroll
    for (int nt = 0; nt < NNT; ++nt) o[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    uint16_t* Vw = Vs[wave];
    auto step = [&](const bf16x8 (&kf)[NKS], const u32x4 (&vf)[VCH], const f32x4& kscl, const f32x4& vscl, int tb, int limit) {
        // stage this tile's V rows in the wave's LDS region
#pragma unroll
        for (int i = 0; i < VCH; ++i) {
            const int c = lane + 64 * i, tok = c / (D / 8), d8 = (c % (D / 8)) * 8;
            *(u32x4*)&Vw[tok * VLD + d8] = vf[i];
        }
        f32x4 sc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            sc = mma_k32<KVT>(kf[ks], qh[ks], sc);
            sc = mma_k32<KVT>(kf[ks], ql[ks], sc);
        }
        float mt = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (KVQ) sc[r] *= kscl[r];
            if (tb + g * 4 + r >= limit) sc[r] = -INFINITY;
            mt = fmaxf(mt, sc[r]);
        }
        mt = fmaxf(mt, __shfl_xor(mt, 16));
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float m_new = fmaxf(m_run, mt);                    // finite: token tb < limit is in this tile
        const float alpha = expf(m_run - m_new);
        float psum = 0.f;
        bf16x4 ph, pl;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float p = expf(sc[r] - m_new);
            psum += p;
            const float pv = KVQ ? p * vscl[r] : p;              // V scale of the token folded into the P^T operand
            const uint16_t hh = kv16_from_f32<KVT>(pv);
            ph[r] = (short)hh;
            pl[r] = (short)kv16_from_f32<KVT>(pv - kv16_to_f32<KVT>(hh));
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int nt = 0; nt < NNT; ++nt) { o[nt][0] *= alpha; o[nt][1] *= alpha; o[nt][2] *= alpha; o[nt][3] *= alpha; }
        __builtin_amdgcn_wave_barrier();                         // LDS tile written by this wave only
#pragma unroll
        for (int nt = 0; nt < NNT; ++nt) {
            const uint16_t* vp = &Vw[(g * 4 + (sub >> 2)) * VLD + nt * 16 + (sub & 3) * 4];
            const bf16x4 vh = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)vp);
            o[nt] = mma_k16<KVT>(vh, ph, o[nt]);
            o[nt] = mma_k16<KVT>(vh, pl, o[nt]);
        }
        __builtin_amdgcn_wave_barrier();                         // tile consumed before the next step overwrites it
    };
    {
        const int stride = TB * nsplit;
        int tb = tb_first;
        if constexpr (D <= 128) {                                // two statically named register sets: next tile in flight
            if (KVQ && tb < pos) load_tile(kA, vA, ksA, vsA, tb);
            while (tb < pos) {
                if (tb + stride < pos) load_tile(kB, vB, ksB, vsB, tb + stride);
                step(kA, vA, ksA, vsA, tb, pos);
                tb += stride;
                if (tb >= pos) break;
                if (tb + stride < pos) load_tile(kA, vA, ksA, vsA, tb + stride);
                step(kB, vB, ksB, vsB, tb, pos);
                tb += stride;
            }
        } else {                                                 // head_dim 256: one register set (the second one spills)
            for (; tb < pos; tb += stride) {
                load_tile(kA, vA, ksA, vsA, tb);
                step(kA, vA, ksA, vsA, tb, pos);
            }
        }
        if (owner && wave == 0) {                                // the token appended by this step: row 0 of a pseudo-tile
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const u32x4 kn = *(const u32x4*)&knew[ks * 32 + g * 8];
                const u32x4 kz = sub == 0 ? kn : (u32x4){0, 0, 0, 0};
                kA[ks] = __builtin_bit_cast(bf16x8, kz);
            }
#pragma unroll
            for (int i = 0; i < VCH; ++i) {
                const int c = lane + 64 * i, tok = c / (D / 8), d8 = (c % (D / 8)) * 8;
                vA[i] = tok == 0 ? *(const u32x4*)&vnew[d8] : (u32x4){0, 0, 0, 0};
            }
            ksA = (f32x4){new_scale[0], 1.f, 1.f, 1.f}; vsA = (f32x4){new_scale[1], 1.f, 1.f, 1.f};
            step(kA, vA, ksA, vsA, pos, pos + 1);
        }
    }
    l_run += __shfl_xor(l_run, 16);
    l_run += __shfl_xor(l_run, 32);

    // ---- merge the 4 waves: O^T rows = dims nt*16 + g*4 + r, column = head `sub` ----
    __syncthreads();
    float* red_o = (float*)&Vs[0][0];                            // [4][NREP][D] f32 (fits: NREP * D * 4 * 4 <= sizeof(Vs))
    if (sub < NREP) {
        if (g == 0) { red_m[wave][sub] = m_run; red_l[wave][sub] = l_run; }
#pragma unroll
        for (int nt = 0; nt < NNT; ++nt)
            *(f32x4*)&red_o[((size_t)wave * NREP + sub) * D + nt * 16 + g * 4] = o[nt];
    }
    __syncthreads();
    for (int it = tid; it < NREP * D; it += 256) {
        const int h = it / D, d = it % D;
        float M = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) M = fmaxf(M, red_m[w][h]);
        float O = 0.f, Ls = 0.f;
        if (M > -INFINITY) {
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float wt = red_m[w][h] > -INFINITY ? expf(red_m[w][h] - M) : 0.f;
                O += wt * red_o[((size_t)w * NREP + h) * D + d];
                Ls += wt * red_l[w][h];
            }
        }
        if (a.out1 != nullptr) {                                 // the only split of this sequence: what the combine kernel would compute
            float v = O * (1.0f / Ls);                           //   from one partial (weight e^0 = 1), bit for bit
            if (a.gate != nullptr) v *= 1.0f / (1.0f + expf(-a.gate[(size_t)bq * a.qkv_stride + (size_t)(kvh * NREP + h) * D + d]));
            if (a.out1_hi != nullptr) {
                const size_t off = (size_t)bq * a.out1_cols + (size_t)(kvh * NREP + h) * D + d;
                const uint16_t hh = f32_to_bf16(v);
                a.out1_hi[off] = hh;
                a.out1_lo[off] = f32_to_bf16(v - bf16_to_f32(hh));
            } else {
                a.out1[(size_t)bq * a.out1_stride + (size_t)(kvh * NREP + h) * D + d] = v;
            }
            if (a.out1_q != nullptr) {
                // the row is the o_proj input of a quantised decode group: 32 consecutive lanes hold one Q8_0 block of it (every wave of
                // the loop is fully active: NREP D is a multiple of 64) -- amax, d = amax / 127, roundf(x / d), f16-rounded scale
                float am = fabsf(v);
                am = fmaxf(am, __shfl_xor(am, 1)); am = fmaxf(am, __shfl_xor(am, 2)); am = fmaxf(am, __shfl_xor(am, 4));
                am = fmaxf(am, __shfl_xor(am, 8)); am = fmaxf(am, __shfl_xor(am, 16));
                const float dq = am / 127.0f;
                const float id = dq != 0.f ? 1.0f / dq : 0.f;
                const int col = (kvh * NREP + h) * D + d;
                a.out1_q[(size_t)bq * a.out1_cols + col] = (signed char)(int)roundf(v * id);
                if ((lane & 31) == 0) {
                    const _Float16 hq = (_Float16)dq;
                    a.out1_qd[(size_t)(col >> 5) * QGEMM_MAXM + bq] = (float)hq;
                }
            }
            continue;
        }
        const size_t ph = ((size_t)bq * a.Hkv * NREP + (size_t)(kvh * NREP + h)) * nsplit + split;
        a.part_o[ph * D + d] = O;
        if (d == 0) { a.part_ml[ph * 2] = M; a.part_ml[ph * 2 + 1] = Ls; }
    }
}

template <int D>
static bool launch_mfma(const AttnDecArgs& a, int nrep, int nsplit, int kv_mode, int n_seq, hipStream_t s) {
    dim3 grid(nsplit, a.Hkv, n_seq), block(256);
#define CM_MF(N) case N: if (kv_mode == 2) hipLaunchKernelGGL((attn_decode_mfma_kernel<D, N, 2>), grid, block, 0, s, a); \
                         else if (kv_mode == 3) hipLaunchKernelGGL((attn_decode_mfma_kernel<D, N, 3>), grid, block, 0, s, a); \
                         else if (kv_mode == KV_F16) hipLaunchKernelGGL((attn_decode_mfma_kernel<D, N, KV_F16>), grid, block, 0, s, a); \
                         else hipLaunchKernelGGL((attn_decode_mfma_kernel<D, N, 0>), grid, block, 0, s, a); return true;
    switch (nrep) {
        CM_MF(1) CM_MF(2) CM_MF(3) CM_MF(4) CM_MF(6) CM_MF(8)
        default: return false;
    }
#undef CM_MF
}

bool attn_decode_single_split(int nsplit, int D) {
    static const int out1_env = getenv("CM_ATTN_OUT1") ? atoi(getenv("CM_ATTN_OUT1")) : 1;
    return nsplit == 1 && out1_env != 0 && (D == 128 || D == 256);
}

// bf16 / f16 / int8 / int4 KV (not f32); same partial format and combine kernel as launch_attn_decode
bool launch_attn_decode_mfma(const AttnDecArgs& a, int D, int nrep, int nsplit, int kv_mode, float* out, int out_stride, int n_seq, hipStream_t s) {
    if (a.page <= 0 || (a.page & (a.page - 1)) != 0 || kv_mode == 1) return false;
    // one token split per sequence (large groups: the (kv head, sequence) pairs alone fill the chip): the kernel writes the
    // normalised, gated output itself -- no combine launch (13 us per layer of a 128-sequence round); CM_ATTN_OUT1 = 0: A/B
    if (attn_decode_single_split(nsplit, D)) {
        AttnDecArgs b = a;
        b.out1 = out; b.out1_stride = out_stride;
        return D == 128 ? launch_mfma<128>(b, nrep, 1, kv_mode, n_seq, s) : launch_mfma<256>(b, nrep, 1, kv_mode, n_seq, s);
    }
    if (a.out1_hi != nullptr || a.out1_q != nullptr) return false;       // (planes / codes are only written by the single-split form)
    if (D == 128) {
        if (!launch_mfma<128>(a, nrep, nsplit, kv_mode, n_seq, s)) return false;
        hipLaunchKernelGGL(attn_decode_combine_kernel<128>, dim3(a.Hkv * nrep, n_seq), dim3(1024), 0, s, a.part_o, a.part_ml,
                           a.gate, out, nsplit, a.qkv_stride, out_stride);
    } else if (D == 256) {
        if (!launch_mfma<256>(a, nrep, nsplit, kv_mode, n_seq, s)) return false;
        hipLaunchKernelGGL(attn_decode_combine_kernel<256>, dim3(a.Hkv * nrep, n_seq), dim3(1024), 0, s, a.part_o, a.part_ml,
                           a.gate, out, nsplit, a.qkv_stride, out_stride);
    } else {
        return false;
    }
    return true;
}

template <int D>
static bool launch_split(const AttnDecArgs& a, int nrep, int nsplit, int kv_mode, int n_seq, hipStream_t s) {
    dim3 grid(nsplit, a.Hkv, n_seq), block(256);
#define CM_ATTN_CASE(N) \
    case N: if (kv_mode == 1) hipLaunchKernelGGL((attn_decode_split_kernel<D, N, 1>), grid, block, 0, s, a); \
            else if (kv_mode == 2) hipLaunchKernelGGL((attn_decode_split_kernel<D, N, 2>), grid, block, 0, s, a); \
            else if (kv_mode == 3) hipLaunchKernelGGL((attn_decode_split_kernel<D, N, 3>), grid, block, 0, s, a); \
            else if (kv_mode == KV_F16) hipLaunchKernelGGL((attn_decode_split_kernel<D, N, KV_F16>), grid, block, 0, s, a); \
            else hipLaunchKernelGGL((attn_decode_split_kernel<D, N, 0>), grid, block, 0, s, a); return true;
    switch (nrep) {
        CM_ATTN_CASE(1) CM_ATTN_CASE(2) CM_ATTN_CASE(3) CM_ATTN_CASE(4) CM_ATTN_CASE(6) CM_ATTN_CASE(8)
        default: return false;
    }
#undef CM_ATTN_CASE
}

bool launch_attn_decode(const AttnDecArgs& a, int D, int nrep, int nsplit, int kv_mode, float* out, int out_stride, int n_seq,
                        hipStream_t s) {
    if (a.out1_q != nullptr) return false;                       // (only the matrix-core kernel quantises its output rows)
    if (attn_decode_single_split(nsplit, D)) {                   // (see launch_attn_decode_mfma)
        AttnDecArgs b = a;
        b.out1 = out; b.out1_stride = out_stride;
        return D == 128 ? launch_split<128>(b, nrep, 1, kv_mode, n_seq, s) : launch_split<256>(b, nrep, 1, kv_mode, n_seq, s);
    }
    if (a.out1_hi != nullptr) return false;
    if (D == 128) {
        if (!launch_split<128>(a, nrep, nsplit, kv_mode, n_seq, s)) return false;
        hipLaunchKernelGGL(attn_decode_combine_kernel<128>, dim3(a.Hkv * nrep, n_seq), dim3(1024), 0, s, a.part_o, a.part_ml,
                           a.gate, out, nsplit, a.qkv_stride, out_stride);
    } else if (D == 256) {
        if (!launch_split<256>(a, nrep, nsplit, kv_mode, n_seq, s)) return false;
        hipLaunchKernelGGL(attn_decode_combine_kernel<256>, dim3(a.Hkv * nrep, n_seq), dim3(1024), 0, s, a.part_o, a.part_ml,
                           a.gate, out, nsplit, a.qkv_stride, out_stride);
    } else {
        return false;
    }
    return true;
}

}  // namespace cm
