// Probe: semantics of ds_read_b64_tr_b16 on gfx950.  LDS image: V[64 tokens][stride] bf16 with
// V[t][d] = t*256 + d (as integer bit patterns).  Lane l supplies the address of
// V[4*(l>>4) + ((l&15)>>2)][(l&3)*4] and we print the 4 values each lane receives.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) short s4;
__global__ void probe(uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[64 * 144];
    for (int i = threadIdx.x; i < 64 * 144; i += 64) { int t = i / 144, d = i % 144; lds[i] = (uint16_t)(t * 256 + d); }
    __syncthreads();
    int l = threadIdx.x, g = l >> 4, sub = l & 15;
    const uint16_t* p = &lds[(4 * g + (sub >> 2)) * 144 + (sub & 3) * 4];
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)p);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)v[j];
}
int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    uint16_t h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) printf(" (t=%d,d=%d)", h[l * 4 + j] / 256, h[l * 4 + j] % 256);
        printf("\n");
    }
    return 0;
}
