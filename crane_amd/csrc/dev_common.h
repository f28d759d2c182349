// Device-side helpers shared by all gfx950 kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cm {

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;

constexpr int WAVE = 64;

// ---- bf16 <-> f32 (bit tricks; RNE on the way down, finite inputs) ---------
__device__ __forceinline__ float bf16_lo(uint32_t packed) { return __uint_as_float(packed << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t packed) { return __uint_as_float(packed & 0xFFFF0000u); }
__device__ __forceinline__ float bf16_to_f32(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
// gfx950 converts in hardware (v_cvt_pk_bf16_f32, round to nearest even): one instruction per PAIR instead of five
// integer ops per value
typedef __attribute__((ext_vector_type(2))) float cm_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 cm_bf16x2;
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const cm_bf16x2 v = __builtin_convertvector((cm_f32x2){lo, hi}, cm_bf16x2);
    uint32_t r;
    __builtin_memcpy(&r, &v, 4);
    return r;
}
__device__ __forceinline__ uint16_t f32_to_bf16(float f) { return (uint16_t)(pack_bf16x2(f, 0.f) & 0xFFFFu); }
__device__ __forceinline__ float bf16_round(float f) { return bf16_to_f32(f32_to_bf16(f)); }

// ---- f16 <-> f32 (CM_KV_F16 pages: K/V cached as IEEE binary16 -- the bytes of a bf16 page, 11 significand bits instead
// of 8; v_cvt_f16_f32 rounds to nearest even and keeps subnormals, the clamp saturates instead of overflowing to inf) ----
typedef __attribute__((ext_vector_type(2))) _Float16 cm_f16x2;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
__device__ __forceinline__ float f16_lo(uint32_t packed) { return (float)__builtin_bit_cast(cm_f16x2, packed)[0]; }
__device__ __forceinline__ float f16_hi(uint32_t packed) { return (float)__builtin_bit_cast(cm_f16x2, packed)[1]; }
__device__ __forceinline__ float f16_to_f32(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ uint16_t f32_to_f16(float f) {
    const _Float16 h = (_Float16)fminf(fmaxf(f, -65504.f), 65504.f);
    return __builtin_bit_cast(uint16_t, h);
}
// internal KV-page element types (template parameter KVT of the attention kernels, Model::kv_mode)
// (KV_BF16X2: not a cache mode -- the vision tower's per-call K/V scratch, every f32 value pre-split into bf16 hi + lo)
enum { KV_BF16 = 0, KV_F32 = 1, KV_INT8 = 2, KV_INT4 = 3, KV_F16 = 4, KV_BF16X2 = 5 };
// one cached 16-bit element pair -> f32 (bf16: bit shifts; f16: v_cvt_f32_f16)
template <int KVT> __device__ __forceinline__ float kv16_lo(uint32_t p) { return KVT == KV_F16 ? f16_lo(p) : bf16_lo(p); }
template <int KVT> __device__ __forceinline__ float kv16_hi(uint32_t p) { return KVT == KV_F16 ? f16_hi(p) : bf16_hi(p); }
template <int KVT> __device__ __forceinline__ uint16_t kv16_from_f32(float f) { return KVT == KV_F16 ? f32_to_f16(f) : f32_to_bf16(f); }
template <int KVT> __device__ __forceinline__ float kv16_to_f32(uint16_t h) { return KVT == KV_F16 ? f16_to_f32(h) : bf16_to_f32(h); }

// KVT = 4 (f16 pages): the same tiles on the f16 matrix-core instructions -- K rows and V^T fragments are the cached halves,
// q and p are split into f16 hi + lo (q is O(10) after the QK-norm, p <= 1: both far inside the binary16 range).
template <int KVT>
__device__ __forceinline__ f32x4 mma_k32(const bf16x8& a, const bf16x8& b, const f32x4& c) {
    if constexpr (KVT == KV_F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <int KVT>
__device__ __forceinline__ f32x4 mma_k16(const bf16x4& a, const bf16x4& b, const f32x4& c) {
    if constexpr (KVT == KV_F16) return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, a), __builtin_bit_cast(f16x4, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}

// ---- streaming (non-temporal) 16-byte load: weights are read once ----------
__device__ __forceinline__ u32x4 ld_nt16(const void* p) {
    return __builtin_nontemporal_load((const u32x4*)p);
}
__device__ __forceinline__ u32x4 ld16(const void* p) { return *(const u32x4*)p; }

// ---- DPP cross-lane (no LDS traffic) ----------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// sum over each aligned group of 16 lanes ("row"); every lane of the row gets it
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);   // row_half_mirror
    v += dpp_mov<0x140>(v);   // row_mirror
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    v = fmaxf(v, dpp_mov<0x140>(v));
    return v;
}
// full 64-lane sum, result wave-uniform
__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (a + b) + (c + d);
}
__device__ __forceinline__ float wave_max(float v) {
    v = row16_max(v);
    float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fmaxf(fmaxf(a, b), fmaxf(c, d));
}

// ---- synthetic weights (crane_amd/synth.py) ------------------------------------
__host__ __device__ __forceinline__ uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
__host__ __device__ __forceinline__ float synth_val(uint32_t idx, uint32_t tseed, float mul, float off) {
#pragma clang fp contract(off)      // (the _rn intrinsics below are plain operators in this ROCm: contraction is switched off by name)
    uint32_t h = fmix32(idx * 0x9E3779B1u + tseed);
    int c = (int)((h & 0xFF) + ((h >> 8) & 0xFF) + ((h >> 16) & 0xFF) + (h >> 24)) - 510;
#if defined(__HIP_DEVICE_COMPILE__)
    return __fadd_rn(off, __fmul_rn((float)c, mul));   // no FMA contraction: must match numpy bit-for-bit
#else
    volatile float prod = (float)c * mul;
    return off + prod;
#endif
}

// decode-step state living in HBM so a captured hipGraph can be replayed
struct StepState {
    uint32_t token;     // token to feed this step
    int32_t pos;        // its KV position
    uint32_t next;      // arg-max result of this step
    int32_t pad;        // token-ring write index
    int32_t slot;       // active sequence slot (indexes the per-sequence GDN state pools)
    int32_t rsv[3];     // rsv[0] = rotary position delta (MRoPE counter - cache position); rsv[1] = epoch base and
                        // rsv[2] = error code of the persistent chain kernel (kernels_engine.hip)
};

}  // namespace cm
