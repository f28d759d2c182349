// Does the hardware f32 atomic add round like a plain f32 add?  (in-place residual via atomicAdd)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k(float* y, const float* b, float* z, const float* a, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { atomicAdd(&y[i], b[i]); z[i] = a[i] + b[i]; }
}
int main() {
    const int n = 1 << 20;
    std::vector<float> a(n), b(n), y(n), z(n);
    srand(1);
    for (int i = 0; i < n; ++i) { a[i] = (rand() / (float)RAND_MAX - 0.5f) * 8.f; b[i] = (rand() / (float)RAND_MAX - 0.5f) * ((i & 1) ? 1e-3f : 2.f); }
    float *da, *db, *dy, *dz;
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dy, n * 4); hipMalloc(&dz, n * 4);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(dy, a.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dy, db, dz, da, n);
    hipMemcpy(y.data(), dy, n * 4, hipMemcpyDeviceToHost); hipMemcpy(z.data(), dz, n * 4, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < n; ++i) if (y[i] != z[i]) ++bad;
    printf("atomic != plain add in %d of %d\n", bad, n);
    return 0;
}
