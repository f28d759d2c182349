// What does a dependent kernel inside a captured graph cost on this box, and which part of a GEMV-shaped kernel is it?
// Chains of NK kernels, each consuming the vector the previous one wrote (x[1024] f32), captured in one hipGraph; prints the
// period per kernel.  (DESIGN 3.6 "fewer launches": the launch-bound configurations sit at ~5 us per dependent kernel.)
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/launch_floor_probe tools/probes/launch_floor_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void k_empty() {}
// every workgroup reads the whole input vector (as the GEMV stages x), reduces it, and writes its 4 outputs
template <int MODE>   // 0: plain loads / stores   1: + `rows` weight rows of 1024 bf16 per wave streamed (non-temporal) before the x read
__global__ __launch_bounds__(256) void k_vec(const float* __restrict__ x, float* __restrict__ y, const unsigned short* __restrict__ W, int rows_per_wave, int K) {
    __shared__ float xs[4096];
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u32x4 q[8];
    float acc = 0.f;
    const int gw = blockIdx.x * 4 + wave;
    if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) q[i] = __builtin_nontemporal_load((const u32x4*)(W + ((size_t)gw * rows_per_wave) * K + i * 512 + lane * 8));
    }
    for (int i = tid; i < K / 4; i += 256) ((f32x4*)xs)[i] = ((const f32x4*)x)[i];
    __syncthreads();
    if (MODE == 1) {
        for (int r = 0; r < rows_per_wave * K / 512 / 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc += __uint_as_float(q[i][0] << 16) * xs[(i * 512 + lane * 8) & (K - 1)] + __uint_as_float(q[i][1] << 16) * xs[(i * 512 + lane * 8 + 2) & (K - 1)]
                     + __uint_as_float(q[i][2] << 16) * xs[(i * 512 + lane * 8 + 4) & (K - 1)] + __uint_as_float(q[i][3] << 16) * xs[(i * 512 + lane * 8 + 6) & (K - 1)];
                if (r + 1 < rows_per_wave * K / 512 / 8)
                    q[i] = __builtin_nontemporal_load((const u32x4*)(W + ((size_t)gw * rows_per_wave) * K + (size_t)(r + 1) * 4096 + i * 512 + lane * 8));
            }
        }
    } else {
        for (int i = lane; i < K; i += 64) acc += xs[i];
    }
    for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (tid < 4) y[(blockIdx.x * 4 + tid) & (K - 1)] = red[tid] * 1e-3f + 1.0f;
}

template <class F> static double chain(const char* name, int NK, F launch, hipStream_t s) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < NK; ++i) launch(i);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%-58s %7.2f us per kernel\n", name, best * 1e3 / NK);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return best * 1e3 / NK;
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    const int K = 1024, NK = 400;
    float *x, *y; unsigned short* W;
    CK(hipMalloc(&x, 4096 * 4)); CK(hipMalloc(&y, 4096 * 4));
    const size_t wbytes = (size_t)512 << 20;
    CK(hipMalloc(&W, wbytes)); CK(hipMemset(W, 0, wbytes)); CK(hipMemset(x, 0, 4096 * 4)); CK(hipMemset(y, 0, 4096 * 4));
    chain("empty <<<1, 64>>>", NK, [&](int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); }, s);
    chain("empty <<<256, 256>>>", NK, [&](int) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s); }, s);
    chain("empty <<<1024, 256>>>", NK, [&](int) { hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, s); }, s);
    for (int grid : {1, 64, 256, 512, 1024}) {
        char nm[96]; snprintf(nm, sizeof nm, "x[1024] -> reduce -> y, <<<%d, 256>>> (no weights)", grid);
        chain(nm, NK, [&](int i) { hipLaunchKernelGGL(k_vec<0>, dim3(grid), dim3(256), 0, s, (i & 1) ? y : x, (i & 1) ? x : y, W, 0, K); }, s);
    }
    // GEMV-shaped: every wave streams nb batches of 8 KB (4 rows of 1024 bf16); a different weight matrix per kernel (8 round robin)
    for (int grid : {128, 256, 512}) for (int nb : {1, 2, 4}) {
        const int rpw = 4 * nb;
        char nm[96]; snprintf(nm, sizeof nm, "GEMV-shaped %.1f MB of bf16 rows, <<<%d, 256>>>", grid * 4.0 * nb * 8192 / 1e6, grid);
        chain(nm, NK, [&](int i) { hipLaunchKernelGGL(k_vec<1>, dim3(grid), dim3(256), 0, s, (i & 1) ? y : x, (i & 1) ? x : y,
                                                       W + (size_t)(i & 7) * ((size_t)32 << 20), rpw, K); }, s);
    }
    // the same without a graph (stream launches)
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < NK; ++i) hipLaunchKernelGGL(k_vec<0>, dim3(256), dim3(256), 0, s, (i & 1) ? y : x, (i & 1) ? x : y, W, 0, K);
            CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("%-58s %7.2f us per kernel\n", "stream launches (no graph), <<<256, 256>>> no weights", ms * 1e3 / NK);
        }
    }
    return 0;
}
