// Utility kernels: deterministic synthetic weights (crane_amd/synth.py), synthetic KV fill.
#include "dev_common.h"
#include "kernels.h"

namespace cm {

__global__ void synth_fill_kernel(uint16_t* __restrict__ dst, size_t dst_row_stride, int nrows, int ncols,
                                  int row0, int col0, int full_cols, uint32_t tseed, float mul, float off) {
    const size_t total = (size_t)nrows * ncols;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / ncols), c = (int)(i % ncols);
        const uint32_t idx = (uint32_t)((size_t)(row0 + r) * full_cols + (col0 + c));
        dst[(size_t)r * dst_row_stride + c] = f32_to_bf16(synth_val(idx, tseed, mul, off));
    }
}

void launch_synth_fill(uint16_t* dst, size_t dst_row_stride, int nrows, int ncols, int row0, int col0,
                       int full_cols, uint32_t tseed, float mul, float off, hipStream_t s) {
    const size_t total = (size_t)nrows * ncols;
    int blocks = (int)std::min<size_t>((total + 255) / 256, 256 * 32);
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(synth_fill_kernel, dim3(blocks), dim3(256), 0, s, dst, dst_row_stride, nrows, ncols,
                       row0, col0, full_cols, tseed, mul, off);
}

// pool: [n_pages][page_elems]; fill the listed pages with N(0,1)-like bf16-representable values
// (values k / 147.8 rounded to bf16: |v| < 3.5 with 8 significand bits, so the f16 page holds the same number exactly)
template <typename T, bool F16>
__global__ void kv_fill_kernel(T* __restrict__ pool, const int32_t* __restrict__ pages, int npages,
                               size_t page_elems, uint32_t tseed, size_t head_elems, int hkv_all, int kvh0) {
    const size_t total = (size_t)npages * page_elems;
    const float mul = 1.0f / 147.80054f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int p = (int)(i / page_elems);
        const size_t e = i % page_elems;
        // the value is a function of the element's index in the UNSHARDED page ([page][all kv heads][token][D]): a tensor-
        // parallel rank (its kv heads kvh0 .. of hkv_all) fills its pages with the values the whole model would hold there
        const size_t gi = ((size_t)p * hkv_all + kvh0 + e / head_elems) * head_elems + e % head_elems;
        const uint16_t b = f32_to_bf16(synth_val((uint32_t)gi, tseed, mul, 0.f));
        if constexpr (sizeof(T) == 2) pool[(size_t)pages[p] * page_elems + e] = F16 ? f32_to_f16(bf16_to_f32(b)) : b;
        else pool[(size_t)pages[p] * page_elems + e] = bf16_to_f32(b);
    }
}

// quantised KV pages (bench set-up): random codes, a fixed plausible per-token scale
__global__ void kv_fill_quant_kernel(uint8_t* __restrict__ pool, const int32_t* __restrict__ pages, int npages, size_t page_bytes,
                                     size_t code_bytes, uint32_t tseed) {
    const size_t words = page_bytes / 4, total = (size_t)npages * words;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int p = (int)(i / words);
        const size_t w = i % words;
        uint32_t v = fmix32((uint32_t)i * 0x9E3779B1u + tseed);
        if (w * 4 >= code_bytes) v = __float_as_uint(0.02f);
        ((uint32_t*)(pool + (size_t)pages[p] * page_bytes))[w] = v;
    }
}
void launch_kv_fill_quant(void* pool, const int32_t* pages, int npages, size_t page_bytes, size_t code_bytes, uint32_t tseed, hipStream_t s) {
    const size_t total = (size_t)npages * (page_bytes / 4);
    int blocks = (int)std::min<size_t>((total + 255) / 256, 256 * 16);
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(kv_fill_quant_kernel, dim3(blocks), dim3(256), 0, s, (uint8_t*)pool, pages, npages, page_bytes, code_bytes, tseed);
}

void launch_kv_fill(void* pool, int kvt, const int32_t* pages, int npages, size_t page_elems, uint32_t tseed,
                    size_t head_elems, int hkv_all, int kvh0, hipStream_t s) {
    const size_t total = (size_t)npages * page_elems;
    int blocks = (int)std::min<size_t>((total + 255) / 256, 256 * 16);
    if (blocks < 1) blocks = 1;
    if (kvt == KV_F32) hipLaunchKernelGGL((kv_fill_kernel<float, false>), dim3(blocks), dim3(256), 0, s, (float*)pool, pages, npages, page_elems, tseed, head_elems, hkv_all, kvh0);
    else if (kvt == KV_F16) hipLaunchKernelGGL((kv_fill_kernel<uint16_t, true>), dim3(blocks), dim3(256), 0, s, (uint16_t*)pool, pages, npages, page_elems, tseed, head_elems, hkv_all, kvh0);
    else hipLaunchKernelGGL((kv_fill_kernel<uint16_t, false>), dim3(blocks), dim3(256), 0, s, (uint16_t*)pool, pages, npages, page_elems, tseed, head_elems, hkv_all, kvh0);
}

}  // namespace cm
