// Probe: arithmetic of v_dot2c_f32_bf16 on gfx950 (D = D + a.lo*b.lo + a.hi*b.hi): which rounding sequence does it follow?
// hipcc --offload-arch=gfx950 -O2 tools/probes/dot2_probe.hip -o tools/probes/dot2_probe && tools/probes/dot2_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const uint32_t* a, const uint32_t* b, const float* c, float* o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    o[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a[i]), __builtin_bit_cast(bf16x2, b[i]), c[i], false);
}
static float bf(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
int main() {
    const int n = 1 << 20;
    std::vector<uint32_t> a(n), b(n); std::vector<float> c(n), o(n);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
    for (int i = 0; i < n; ++i) {
        auto mk = [&](int spread) { uint32_t r = rnd(); uint16_t e = 127 - spread / 2 + (r >> 8) % (spread + 1); return (uint16_t)(((r & 1) << 15) | (e << 7) | ((r >> 16) & 0x7F)); };
        a[i] = mk(8) | ((uint32_t)mk(8) << 16); b[i] = mk(8) | ((uint32_t)mk(8) << 16);
        uint32_t r = rnd(); uint32_t e = 127 - 6 + (r >> 8) % 12; uint32_t u = ((r & 1) << 31) | (e << 23) | (rnd() & 0x7FFFFF); memcpy(&c[i], &u, 4);
        if (i % 16 == 0) c[i] = 0.f;
    }
    uint32_t *da, *db; float *dc, *d_o;
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&d_o, n * 4);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(da, db, dc, d_o, n);
    hipMemcpy(o.data(), d_o, n * 4, hipMemcpyDeviceToHost);
    long m_exact = 0, m_seq01 = 0, m_seq10 = 0, m_prodfirst = 0; double maxrel = 0;
    for (int i = 0; i < n; ++i) {
        const float a0 = bf(a[i] & 0xFFFF), a1 = bf(a[i] >> 16), b0 = bf(b[i] & 0xFFFF), b1 = bf(b[i] >> 16);
        const double ex = (double)c[i] + (double)a0 * b0 + (double)a1 * b1;
        const float r_exact = (float)ex;                                   // one rounding
        const float r01 = fmaf(a1, b1, fmaf(a0, b0, c[i]));                 // c + p0, then + p1
        const float r10 = fmaf(a0, b0, fmaf(a1, b1, c[i]));
        const float rp = (a0 * b0 + a1 * b1) + c[i];                        // products summed first (p exact in f32), then + c
        m_exact += o[i] == r_exact; m_seq01 += o[i] == r01; m_seq10 += o[i] == r10; m_prodfirst += o[i] == rp;
        if (ex != 0) { double r = fabs((double)o[i] - ex) / fabs(ex); if (r > maxrel && fabs(ex) > 1e-3) maxrel = r; }
    }
    printf("n=%d  == single rounding: %ld  == fma(p1, fma(p0, c)): %ld  == fma(p0, fma(p1, c)): %ld  == (p0+p1)+c: %ld  max rel err %.3e\n", n, m_exact, m_seq01, m_seq10, m_prodfirst, maxrel);
    // denormal inputs / results
    return 0;
}
