// Probe: how fast can 256 workgroups stream a [N, K] bf16 matrix when every wave-instruction fetches 8 rows x 128 B (rows K * 2 bytes
// apart: the access pattern of a GEMM weight tile walked in 64-element k-tiles) against the same bytes laid out tile-major
// (every instruction 1 KB contiguous)?   hipcc --offload-arch=gfx950 -O3 tilestream_probe.hip -o tilestream_probe && ./tilestream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void stream_kernel(const unsigned short* __restrict__ W, int N, int K, int BN, int ksplit, unsigned int* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tiles_n = N / BN, ks = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int nk = K / 64 / ksplit, k0 = ks * nk;
    const int pieces = BN / 8 / 8;                       // 8-row pieces per wave
    u32x4 acc = {0, 0, 0, 0};
    u32x4 r[DEPTH][4];
    auto load = [&](int t, u32x4 (&q)[4]) {
        const int tt = min(t, nk - 1) + k0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i >= pieces) { q[i] = (u32x4){0, 0, 0, 0}; continue; }
            const int p = wave + 8 * i;
            size_t off;
            if (MODE == 0) off = ((size_t)(tn * BN + p * 8 + (lane >> 3)) * K + (size_t)tt * 64 + (lane & 7) * 8);          // row-major tile rows
            else off = (((size_t)tn * (K / 64) + tt) * BN * 64 + (size_t)p * 512 + lane * 8);                              // tile-major: contiguous
            q[i] = __builtin_nontemporal_load((const u32x4*)(W + off));
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) load(d, r[d]);
    for (int t = 0; t < nk; t += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc += r[d][i];
            load(t + d + DEPTH, r[d]);
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 0x12345678u) sink[0] = 1;
}
int main() {
    const int N = 24576, K = 4096;
    unsigned short* W; unsigned int* sink;
    hipMalloc(&W, (size_t)N * K * 2 + (1 << 20)); hipMalloc(&sink, 4);
    hipMemset(W, 1, (size_t)N * K * 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto kern, const char* name, int BN, int ksplit) {
        const int blocks = N / BN * ksplit;
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, W, N, K, BN, ksplit, sink);
        hipEventRecord(e0);
        for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, W, N, K, BN, ksplit, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s BN %3d ksplit %d blocks %4d: %7.1f us  %6.2f TB/s\n", name, BN, ksplit, blocks, ms * 50, (double)N * K * 2 / (ms / 20 * 1e-3) / 1e12);
    };
    for (int ks : {1, 2}) for (int bn : {256, 192}) {
        run(stream_kernel<0, 2>, "rows 8 KB apart, 128 B each, 2 tiles in flight", bn, ks);
        run(stream_kernel<0, 4>, "rows 8 KB apart, 128 B each, 4 tiles in flight", bn, ks);
        run(stream_kernel<1, 2>, "tile-major (1 KB contiguous), 2 in flight", bn, ks);
        run(stream_kernel<1, 4>, "tile-major (1 KB contiguous), 4 in flight", bn, ks);
    }
    return 0;
}
