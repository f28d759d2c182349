// Vision-tower (Qwen 3.5-VL ViT) row kernels; the projections reuse the MFMA GEMM of
// kernels_prefill.hip (bias / GELU epilogues) and the bidirectional per-frame attention reuses
// attn_prefill_kernel<64> in window mode.  Reference: crane-core/src/models/qwen3_5/vision.rs.
#include <cstdlib>

#include "dev_common.h"
#include "kernels.h"

namespace cm {

__device__ __forceinline__ void split_store4v(uint16_t* hi, uint16_t* lo, size_t off, const float v[4]) {
    uint16_t h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { h[i] = f32_to_bf16(v[i]); l[i] = f32_to_bf16(v[i] - bf16_to_f32(h[i])); }
    *(u32x2*)(hi + off) = (u32x2){(uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16)};
    if (lo) *(u32x2*)(lo + off) = (u32x2){(uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16)};
}

// LayerNorm with bias, eps 1e-6 (vision.rs:195-203,250-255): one block per row -> bf16 hi (+lo)
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ b, uint16_t* __restrict__ hi,
                                                             uint16_t* __restrict__ lo, int H, float eps) {
    __shared__ float red[8];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* xr = x + (size_t)row * H;
    float s = 0.f;
    for (int i = tid * 4; i < H; i += 1024) { const f32x4 v = *(const f32x4*)(xr + i); s += v[0] + v[1] + v[2] + v[3]; }
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float mu = ((red[0] + red[1]) + (red[2] + red[3])) / (float)H;
    float q = 0.f;
    for (int i = tid * 4; i < H; i += 1024) {
        const f32x4 v = *(const f32x4*)(xr + i);
        q += (v[0] - mu) * (v[0] - mu) + (v[1] - mu) * (v[1] - mu) + (v[2] - mu) * (v[2] - mu) + (v[3] - mu) * (v[3] - mu);
    }
    q = wave_sum(q);
    if (lane == 0) red[4 + wave] = q;
    __syncthreads();
    const float r = 1.0f / sqrtf(((red[4] + red[5]) + (red[6] + red[7])) / (float)H + eps);
    for (int i = tid * 4; i < H; i += 1024) {
        const f32x4 v = *(const f32x4*)(xr + i);
        const f32x4 ww = *(const f32x4*)(w + i), bb = *(const f32x4*)(b + i);
        const float o[4] = {(v[0] - mu) * r * ww[0] + bb[0], (v[1] - mu) * r * ww[1] + bb[1],
                            (v[2] - mu) * r * ww[2] + bb[2], (v[3] - mu) * r * ww[3] + bb[3]};
        split_store4v(hi, lo, (size_t)row * H + i, o);
    }
}

// x[n, :] += sum_k wts[k][n] * table[idx[k][n], :]   (fast_pos_embed_interpolate, vision.rs:382-489; the
// block-major permutation is folded into idx/wts on the host)
__global__ void pos_embed_add_kernel(float* __restrict__ x, const uint16_t* __restrict__ table, const int32_t* __restrict__ idx,
                                     const float* __restrict__ wts, int N, int H) {
    const int n = blockIdx.x;
    for (int i = threadIdx.x; i < H; i += blockDim.x) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) acc += wts[k * N + n] * bf16_to_f32(table[(size_t)idx[k * N + n] * H + i]);
        x[(size_t)n * H + i] += acc;
    }
}

// grid (N, 3 * heads), block 64 (one lane per head dim, hd == 64): q/k get q*cos + rotate_half(q)*sin over the
// whole head (vision.rs:86-106); q scaled by 1/sqrt(hd) -> bf16 hi/lo [N, heads, hd]; k, v -> scratch in the paged layout
// [page][heads][64 tokens][hd] with an identity block table, every f32 value split ONCE here into bf16 hi + lo (two arrays
// lo_off elements apart): the attention kernel reads each K/V row from ceil(N / 64) query blocks and used to redo the
// split (3 conversions per element) in every one of them -- it was VALU-bound on that.
__global__ __launch_bounds__(64) void vit_rope_kv_kernel(const float* __restrict__ qkv, const float* __restrict__ cs,
                                                         const float* __restrict__ sn, uint16_t* __restrict__ q_hi,
                                                         uint16_t* __restrict__ q_lo, uint16_t* __restrict__ kpool,
                                                         uint16_t* __restrict__ vpool, size_t lo_off, int heads, float scale) {
    constexpr int HD = 64;
    const int n = blockIdx.x, item = blockIdx.y, d = threadIdx.x;
    const int which = item / heads, h = item % heads;            // 0 q, 1 k, 2 v  (reshape (N, 3, heads, hd))
    const float x = qkv[(size_t)n * 3 * heads * HD + (size_t)item * HD + d];
    float o = x;
    if (which < 2) {
        const float partner = __shfl_xor(x, 32);                 // rotate_half: d < 32 -> -x[d+32], else x[d-32]
        const float rh = d < 32 ? -partner : partner;
        o = x * cs[(size_t)n * HD + d] + rh * sn[(size_t)n * HD + d];
    }
    if (which == 0) {
        o *= scale;
        const size_t off = ((size_t)n * heads + h) * HD + d;
        const uint16_t hh = f32_to_bf16(o);
        q_hi[off] = hh; q_lo[off] = f32_to_bf16(o - bf16_to_f32(hh));
    } else {
        uint16_t* pool = which == 1 ? kpool : vpool;
        const size_t off = ((size_t)((n >> 6) * heads + h) * 64 + (n & 63)) * HD + d;
        const uint16_t hh = f32_to_bf16(o);
        pool[off] = hh; pool[lo_off + off] = f32_to_bf16(o - bf16_to_f32(hh));
    }
}

// The same, four elements per thread (round 6): the one-element kernel above was 37 632 workgroups of 64 threads with 4-byte loads and
// 2-byte stores for 784 patches (10.4 us per block of the tower); here a thread owns 4 consecutive d of a head (16 lanes = one head, the
// rotate-half partner is 8 lanes away), 256 threads cover 1024 elements of a row.  Per element the same expression.
__global__ __launch_bounds__(256) void vit_rope_kv4_kernel(const float* __restrict__ qkv, const float* __restrict__ cs,
                                                           const float* __restrict__ sn, uint16_t* __restrict__ q_hi,
                                                           uint16_t* __restrict__ q_lo, uint16_t* __restrict__ kpool,
                                                           uint16_t* __restrict__ vpool, size_t lo_off, int heads, float scale) {
    constexpr int HD = 64;
    const int n = blockIdx.x;
    const int e0 = ((int)blockIdx.y * 256 + (int)threadIdx.x) * 4;      // element of the row (N, 3, heads, hd)
    if (e0 >= 3 * heads * HD) return;                                    // (whole heads: 16 lanes leave together)
    const int item = e0 / HD, d0 = e0 % HD;
    const int which = item / heads, h = item % heads;
    const f32x4 x = *(const f32x4*)(qkv + (size_t)n * 3 * heads * HD + e0);
    f32x4 o = x;
    if (which < 2) {
        const f32x4 c4 = *(const f32x4*)(cs + (size_t)n * HD + d0), s4 = *(const f32x4*)(sn + (size_t)n * HD + d0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float partner = __shfl_xor(x[e], 8);                   // rotate_half: d < 32 -> -x[d+32], else x[d-32]
            const float rh = d0 < 32 ? -partner : partner;
            o[e] = x[e] * c4[e] + rh * s4[e];
        }
    }
    uint32_t hi2[2], lo2[2];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float v = which == 0 ? o[e] * scale : o[e];
        const uint16_t hh = f32_to_bf16(v), ll = f32_to_bf16(v - bf16_to_f32(hh));
        if (e & 1) { hi2[e >> 1] |= (uint32_t)hh << 16; lo2[e >> 1] |= (uint32_t)ll << 16; }
        else { hi2[e >> 1] = hh; lo2[e >> 1] = ll; }
    }
    if (which == 0) {
        const size_t off = ((size_t)n * heads + h) * HD + d0;
        *(u32x2*)(q_hi + off) = (u32x2){hi2[0], hi2[1]};
        *(u32x2*)(q_lo + off) = (u32x2){lo2[0], lo2[1]};
    } else {
        uint16_t* pool = which == 1 ? kpool : vpool;
        const size_t off = ((size_t)((n >> 6) * heads + h) * 64 + (n & 63)) * HD + d0;
        *(u32x2*)(pool + off) = (u32x2){hi2[0], hi2[1]};
        *(u32x2*)(pool + lo_off + off) = (u32x2){lo2[0], lo2[1]};
    }
}

// dst[s, :] = src[map[s], :] where map[s] >= 0  (splice_image_features, vlm.rs:433-468)
__global__ void splice_rows_kernel(float* __restrict__ dst, const float* __restrict__ src, const int32_t* __restrict__ map, int H) {
    const int s = blockIdx.x;
    const int m = map[s];
    if (m < 0) return;
    for (int i = threadIdx.x * 4; i < H; i += blockDim.x * 4) *(f32x4*)(dst + (size_t)s * H + i) = *(const f32x4*)(src + (size_t)m * H + i);
}

void launch_layernorm_rows(const float* x, const float* w, const float* b, uint16_t* hi, uint16_t* lo, int N, int H, float eps,
                           hipStream_t s) {
    hipLaunchKernelGGL(layernorm_rows_kernel, dim3(N), dim3(256), 0, s, x, w, b, hi, lo, H, eps);
}
void launch_pos_embed_add(float* x, const uint16_t* table, const int32_t* idx, const float* wts, int N, int H, hipStream_t s) {
    hipLaunchKernelGGL(pos_embed_add_kernel, dim3(N), dim3(256), 0, s, x, table, idx, wts, N, H);
}
void launch_vit_rope_kv(const float* qkv, const float* cs, const float* sn, uint16_t* q_hi, uint16_t* q_lo, uint16_t* kpool,
                        uint16_t* vpool, size_t lo_off, int N, int heads, float scale, hipStream_t s) {
    static const int v4 = getenv("CM_VIT_ROPE4") ? atoi(getenv("CM_VIT_ROPE4")) : 1;          // 0: the one-element kernel (A/B)
    if (v4 && (lo_off % 4) == 0) {
        hipLaunchKernelGGL(vit_rope_kv4_kernel, dim3(N, (3 * heads * 64 + 1023) / 1024), dim3(256), 0, s, qkv, cs, sn, q_hi, q_lo, kpool, vpool, lo_off, heads, scale);
        return;
    }
    hipLaunchKernelGGL(vit_rope_kv_kernel, dim3(N, 3 * heads), dim3(64), 0, s, qkv, cs, sn, q_hi, q_lo, kpool, vpool, lo_off, heads, scale);
}
// DeepStack injection (qwen3_vl/text.rs:280-333): dst[s, :] += src[map[s], :] for the visual positions (map[s] >= 0)
__global__ __launch_bounds__(256) void add_rows_map_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                           const int32_t* __restrict__ map, int H) {
    const int r = map[blockIdx.x];
    if (r < 0) return;
    float* d = dst + (size_t)blockIdx.x * H;
    const float* v = src + (size_t)r * H;
    for (int i = threadIdx.x * 4; i < H; i += 1024) {
        const f32x4 a = *(const f32x4*)(d + i), b = *(const f32x4*)(v + i);
        *(f32x4*)(d + i) = (f32x4){a[0] + b[0], a[1] + b[1], a[2] + b[2], a[3] + b[3]};
    }
}
void launch_add_rows_map(float* dst, const float* src, const int32_t* map, int S, int H, hipStream_t s) {
    hipLaunchKernelGGL(add_rows_map_kernel, dim3(S), dim3(256), 0, s, dst, src, map, H);
}
void launch_splice_rows(float* dst, const float* src, const int32_t* map, int S, int H, hipStream_t s) {
    hipLaunchKernelGGL(splice_rows_kernel, dim3(S), dim3(256), 0, s, dst, src, map, H);
}

}  // namespace cm
