/*
 * ggml's Q8_0 / Q4_0 / Q5_0 reference quantisers and ggml_vec_dot_q8_0_q8_0, restated in C for the parity tests of the decode groups
 * on the int8 matrix cores (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).
 *
 * What the reference computes here: crane-core/src/ops/linear.rs:18-51 (`LinearLayer::Quantized`: candle's CPU `QMatMul`, whose
 * forward quantises the f32 activation row to the weight type's VecDotType and calls the type's `vec_dot`) and ops/linear.rs:53-116
 * (in-situ quantisation through `QTensor::quantize`).  candle-core 0.11 (`quantized/k_quants.rs`, a port of ggml-quants.c) is a
 * crates.io dependency that is NOT under /root/reference (no lockfile): the published ggml algorithm is restated --
 *   quantize_row_q8_0_ref : d = amax / 127 stored as f16, id = d ? 1 / d : 0, q = roundf(x * id)
 *   quantize_row_q4_0_ref : d = max / -8 (max = signed value of the first element of largest magnitude), q = MIN(15, (int8)(x * id + 8.5f))
 *   quantize_row_q5_0_ref : d = max / -16, q = MIN(31, (int8)(x * id + 16.5f))
 *   ggml_vec_dot_q8_0_q8_0: sumf += sumi * (fp16_to_fp32(x[ib].d) * fp16_to_fp32(y[ib].d)), blocks in ascending order
 *   (ggml_vec_dot_q4_0_q8_0 / _q5_0_q8_0 are the same sum over the codes q - 8 / q - 16: they fit an int8 under the block's own scale)
 * -- PARITY UNPINNED against candle / ggml themselves (neither is in this image); pinned on the numpy restatement
 * oracle/gguf_oracle.py (bit-equal, tests/test_gguf_oracle.py), which carries the hand-built known-answer blocks.
 *
 * Also here: the synthetic-weight generator as an f32 tensor (crane_amd/synth.py synth_weights_f32, bit-identical; numpy needs 75 s
 * for the 1.6 G elements of a 2-layer model at the Qwen3-8B widths, this 1 s).
 */
#include "qc_common.h"

/* out[r * cols + c] = bf16-rounded synthetic value of tensor `name` (std, off: crane_amd/synth.py specs), as f32 */
void qc_synth_f32(const char* name, uint64_t seed, double std, float off, int rows, int cols, float* out) {
    const uint32_t ts = fmix32(fnv1a32(name) ^ (uint32_t)((uint32_t)seed * 0x85EBCA6Bu + 0x1234567u));
    const float mul = (float)(std / sqrt(21845.0));
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) {
            const uint32_t idx = (uint32_t)((size_t)r * cols + c);
            const uint32_t h = fmix32(idx * 0x9E3779B1u + ts);
            const int k = (int)((h & 0xFF) + ((h >> 8) & 0xFF) + ((h >> 16) & 0xFF) + (h >> 24)) - 510;
            out[(size_t)r * cols + c] = bf2f(f2bf(off + (float)k * mul));
        }
}

/* x[n] (n % 32 == 0) -> codes[n] int8 and d[n / 32] = the block scale after its f16 round trip.
 * fmt 8: Q8_0; 2: Q4_0 (codes q - 8 in [-8, 7]); 6: Q5_0 (codes q - 16 in [-16, 15]) */
int qc_quantize_ref(int fmt, const float* x, size_t n, int8_t* codes, float* d_out) {
    if (n % 32 != 0 || (fmt != 8 && fmt != 2 && fmt != 6)) return -1;
    const size_t nb = n / 32;
#pragma omp parallel for schedule(static)
    for (size_t b = 0; b < nb; ++b) {
        const float* xb = x + b * 32;
        int8_t* qb = codes + b * 32;
        if (fmt == 8) {
            float amax = 0.f;
            for (int j = 0; j < 32; ++j) { const float v = fabsf(xb[j]); if (v > amax) amax = v; }
            const float d = amax / 127.0f;
            const float id = d != 0.f ? 1.0f / d : 0.f;
            for (int j = 0; j < 32; ++j) qb[j] = (int8_t)roundf(xb[j] * id);
            d_out[b] = f16_round(d);
        } else {
            const float top = fmt == 2 ? 8.f : 16.f;
            float amax = 0.f, max = 0.f;
            for (int j = 0; j < 32; ++j) { const float v = xb[j]; if (amax < fabsf(v)) { amax = fabsf(v); max = v; } }
            const float d = max / -top;
            const float id = d != 0.f ? 1.0f / d : 0.f;
            for (int j = 0; j < 32; ++j) {
                const float t = xb[j] * id + (top + 0.5f);          /* (-ffp-contract=off: product, then sum, like the C reference) */
                int q = (int)(int8_t)t;
                if (q > (int)(2 * top - 1)) q = (int)(2 * top - 1);
                qb[j] = (int8_t)(q - (int)top);
            }
            d_out[b] = f16_round(d);
        }
    }
    return 0;
}

/* out[m * N + n] = ggml_vec_dot_q8_0_q8_0(row n of the weights, activation row m): wq[N][K] int8 codes, wd[N][K / 32] block scales,
 * xq[M][K], xd[M][K / 32]; f32 accumulation over the blocks in ascending order, like the C reference. */
int qc_vec_dot_q8_rows(const int8_t* wq, const float* wd, int N, int K, const int8_t* xq, const float* xd, int M, float* out) {
    if (K % 32 != 0) return -1;
    const int nb = K / 32;
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n) {
        const int8_t* wr = wq + (size_t)n * K;
        const float* wdr = wd + (size_t)n * nb;
        for (int m = 0; m < M; ++m) {
            const int8_t* xr = xq + (size_t)m * K;
            const float* xdr = xd + (size_t)m * nb;
            float sumf = 0.f;
            for (int ib = 0; ib < nb; ++ib) {
                int sumi = 0;
                for (int j = 0; j < 32; ++j) sumi += (int)wr[ib * 32 + j] * (int)xr[ib * 32 + j];
                sumf += (float)sumi * (wdr[ib] * xdr[ib]);
            }
            out[(size_t)m * N + n] = sumf;
        }
    }
    return 0;
}
