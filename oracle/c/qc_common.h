/* Helpers shared by the two CPU ports (qwen3_cpu.c: dense Qwen3; qwen35_cpu.c: Qwen 3.5 hybrid).  TEST INFRASTRUCTURE ONLY --
 * see oracle/__init__.py.  Deterministic synthetic weights: the generator of crane_amd/synth.py, bit-identical. */
#ifndef QC_COMMON_H
#define QC_COMMON_H
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline float bf2f(uint16_t b) { uint32_t u = ((uint32_t)b) << 16; float f; memcpy(&f, &u, 4); return f; }
static inline uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16); }

/* round to IEEE binary16 and back (RNE, saturating at +-65504, subnormals kept): the rounding point of the device's
 * CM_KV_F16 pages (v_cvt_f16_f32 + clamp).  kv_bf16 == 2 selects it. */
static inline float f16_round(float f) {
    if (f > 65504.f) f = 65504.f;
    if (f < -65504.f) f = -65504.f;
    uint32_t u; memcpy(&u, &f, 4);
    const uint32_t sign = u & 0x80000000u;
    uint32_t a = u & 0x7FFFFFFFu;
    float r;
    if (a >= 0x38800000u) {                     /* |f| >= 2^-14: normal half, keep 10 mantissa bits (RNE on bit 13) */
        a = (a + 0xFFFu + ((a >> 13) & 1u)) & ~0x1FFFu;
        memcpy(&r, &a, 4);
        if (r > 65504.f) r = 65504.f;
    } else {                                    /* subnormal half: multiples of 2^-24 */
        float m; memcpy(&m, &a, 4);
        r = rintf(m * 16777216.0f) * (1.0f / 16777216.0f);   /* rintf = RNE under the default rounding mode */
    }
    uint32_t o; memcpy(&o, &r, 4); o |= sign; memcpy(&r, &o, 4);
    return r;
}

static uint32_t fnv1a32(const char* s) { uint32_t h = 0x811C9DC5u; for (; *s; ++s) { h ^= (unsigned char)*s; h *= 0x01000193u; } return h; }
static inline uint32_t fmix32(uint32_t h) { h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16; return h; }

/* dst[r*stride + c] = synth(name)[(row0+r)*full_cols + c] */
static void synth_rows(uint16_t* dst, size_t stride, const char* name, uint64_t seed, double std, float off,
                       int row0, int nrows, int full_cols) {
    const uint32_t ts = fmix32(fnv1a32(name) ^ (uint32_t)((uint32_t)seed * 0x85EBCA6Bu + 0x1234567u));
    const float mul = (float)(std / sqrt(21845.0));
#pragma omp parallel for schedule(static)
    for (int r = 0; r < nrows; ++r)
        for (int c = 0; c < full_cols; ++c) {
            const uint32_t idx = (uint32_t)((size_t)(row0 + r) * full_cols + c);
            const uint32_t h = fmix32(idx * 0x9E3779B1u + ts);
            const int k = (int)((h & 0xFF) + ((h >> 8) & 0xFF) + ((h >> 16) & 0xFF) + (h >> 24)) - 510;
            const float prod = (float)k * mul;
            dst[(size_t)r * stride + c] = f2bf(off + prod);
        }
}

static void* xmalloc(size_t n) { void* p = malloc(n ? n : 1); if (!p) { fprintf(stderr, "qwen3_cpu: out of memory (%zu)\n", n); abort(); } return p; }

/* y[n] = W[n,:] . x  (bf16 weights, f32 accumulate), rows split over the host cores.
 * 16 independent lane accumulators (lane j sums the products of k = j mod 16 in ascending k, then a fixed tree): written
 * with GCC vector extensions so the bf16 -> f32 widening and the multiply/add are SIMD (AVX2 / AVX-512 with -march=native)
 * without re-associating anything -- bit-identical to the scalar loop it replaces (-ffp-contract=off: no FMA). */
typedef float v16f __attribute__((vector_size(64)));
typedef uint16_t v16h __attribute__((vector_size(32)));
typedef uint32_t v16u __attribute__((vector_size(64)));

static void gemv(const uint16_t* W, const float* x, float* y, int N, int K) {
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n) {
        const uint16_t* w = W + (size_t)n * K;
        v16f acc = {0};
        int k = 0;
        for (; k + 16 <= K; k += 16) {
            v16h h;
            v16f xv, f;
            memcpy(&h, w + k, sizeof h);
            memcpy(&xv, x + k, sizeof xv);
            const v16u u = __builtin_convertvector(h, v16u) << 16;
            memcpy(&f, &u, sizeof f);
            acc += f * xv;
        }
        float a[16];
        memcpy(a, &acc, sizeof a);
        for (; k < K; ++k) a[k & 15] += bf2f(w[k]) * x[k];
        float s8[8];
        for (int j = 0; j < 8; ++j) s8[j] = a[j] + a[j + 8];
        y[n] = ((s8[0] + s8[4]) + (s8[1] + s8[5])) + ((s8[2] + s8[6]) + (s8[3] + s8[7]));
    }
}

#endif
