// Latency of a cross-workgroup hand-off through an 8-byte {value, tag} granule (kernels_engine.hip), by cache-scope bits of the store
// and of the polling load, between workgroups on the SAME XCD (blockIdx % 8 equal) and on different XCDs.  Ping-pong: workgroup A
// stores tag i, workgroup B polls for it and stores its own granule with tag i, A polls for that: round trip / 2 = one hop.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/hop_probe tools/probes/hop_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef unsigned long long u64;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int B> __device__ __forceinline__ u64 ld(const u64* p) {
    u32x2 v;
    if (B == 0) asm volatile("global_load_dwordx2 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (B == 1) asm volatile("global_load_dwordx2 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (B == 2) asm volatile("global_load_dwordx2 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (B == 3) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return ((u64)v[1] << 32) | v[0];
}
template <int B> __device__ __forceinline__ void st(u64* p, u64 x) {
    u32x2 v = {(unsigned)x, (unsigned)(x >> 32)};
    if (B == 0) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(v) : "memory");
    if (B == 1) asm volatile("global_store_dwordx2 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
    if (B == 2) asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    if (B == 3) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
}

// workgroup `wa` and workgroup `wb` play; everybody else exits (or streams `W` when load != 0)
template <int SB, int LB>
__global__ __launch_bounds__(64) void pingpong(u64* ga, u64* gb, int wa, int wb, int iters, unsigned tag0, u64* out, const u32x4* W, size_t wn, int load) {
    const int b = blockIdx.x;
    if (b != wa && b != wb) {
        if (load) {      // background weight stream: every other workgroup reads its slice of W over and over while the two play
            u32x4 acc = {0, 0, 0, 0};
            const size_t per = wn / gridDim.x;
            for (int rep = 0; rep < load; ++rep)
                for (size_t i = threadIdx.x; i < per; i += 64) { u32x4 v = __builtin_nontemporal_load(W + (size_t)b * per + i); acc ^= v; }
            if (acc[0] == 0x12345u) out[8] = acc[1];
        }
        return;
    }
    const int lane = threadIdx.x;
    u64 t0 = __builtin_amdgcn_s_memrealtime();
    unsigned fails = 0;
    for (int i = 1; i <= iters; ++i) {
        const unsigned tag = tag0 + i;
        if (b == wa) st<SB>(ga + lane, ((u64)tag << 32) | lane);
        const u64* src = (b == wa) ? gb : ga;
        unsigned spins = 0;
        for (;;) {
            const u64 x = ld<LB>(src + lane);
            if (__all((unsigned)(x >> 32) == tag)) break;
            if (++spins > 20000u) { ++fails; break; }
        }
        if (b == wb) st<SB>(gb + lane, ((u64)tag << 32) | lane);
        if (fails) break;
    }
    u64 t1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) { out[b == wa ? 0 : 2] = t1 - t0; out[b == wa ? 1 : 3] = fails; }
}

template <int SB, int LB> static void run(const char* nm, int wa, int wb, u64* ga, u64* gb, u64* out, unsigned& tag, const u32x4* W, size_t wn, int load) {
    const int iters = 200;
    CK(hipMemset(out, 0, 64));
    hipLaunchKernelGGL((pingpong<SB, LB>), dim3(256), dim3(64), 0, 0, ga, gb, wa, wb, iters, tag, out, W, wn, load);
    CK(hipDeviceSynchronize());
    tag += iters + 8;
    u64 h[4]; CK(hipMemcpy(h, out, 32, hipMemcpyDeviceToHost));
    if (h[1] || h[3]) printf("%-44s wg %3d <-> %3d  %s: NEVER ARRIVED (stale cache line)\n", nm, wa, wb, load ? "loaded" : "idle  ");
    else printf("%-44s wg %3d <-> %3d  %s: %6.2f us per hop\n", nm, wa, wb, load ? "loaded" : "idle  ", (double)h[0] * 0.01 / iters / 2);
}


// ---- part 2: the two playing workgroups carry their OWN weight stream (4 more waves with 32 x 16-byte loads per lane in flight, as the
// stream waves of engine_kernel do): a poll of the ping-pong wave then queues behind its CU's burst in the vector-memory path.  Variants of
// the poll / the store: vector (sc1) or SCALAR (s_load / s_store glc through the scalar cache: a different path out of the CU).
__device__ __forceinline__ u64 sld(const u64* p) {
    u64 v;
    asm volatile("s_dcache_inv\n s_load_dwordx2 %0, %1, 0x0 glc\n s_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}
__device__ __forceinline__ void sst(u64* p, u64 x) {
    asm volatile("s_store_dwordx2 %0, %1, 0x0 glc\n s_dcache_wb\n s_waitcnt lgkmcnt(0)" :: "s"(x), "s"(p) : "memory");
}
template <int MODE>      // 0: vector store / vector load (sc1)   1: vector store sc1 / scalar load   2: scalar store / scalar load   3: scalar store / vector load sc1
__global__ __launch_bounds__(320) void pingpong_own(u64* ga, u64* gb, int wa, int wb, int iters, unsigned tag0, u64* out, const u32x4* W, size_t wn, int streamers) {
    __shared__ unsigned done;
    const int b = blockIdx.x;
    if (b != wa && b != wb) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    if (wave > 0) {
        if (wave > streamers) return;
        // streamer: 4 sets of 8 loads per lane in flight, forever (until the ping-pong wave is done)
        u32x4 q[4][8];
        u32x4 acc = {0, 0, 0, 0};
        const size_t span = wn / 8;                 // this wave's slice (u32x4 elements)
        const u32x4* base = W + (size_t)((b == wa ? 0 : 4) + wave - 1) * span;
        size_t off = lane;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) { q[j][i] = __builtin_nontemporal_load(base + off); off = (off + 64) % span; }
        while (__hip_atomic_load(&done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { acc ^= q[j][i]; q[j][i] = __builtin_nontemporal_load(base + off); off = (off + 64) % span; }
            }
        }
        if (acc[0] == 0x12345u) out[8] = acc[1];
        return;
    }
    const u64* src = (b == wa) ? gb : ga;
    u64* dst = (b == wa) ? ga : gb;
    // uniform pointers for the scalar path
    const unsigned slo = __builtin_amdgcn_readfirstlane((unsigned)(size_t)src), shi = __builtin_amdgcn_readfirstlane((unsigned)((size_t)src >> 32));
    const unsigned dlo = __builtin_amdgcn_readfirstlane((unsigned)(size_t)dst), dhi = __builtin_amdgcn_readfirstlane((unsigned)((size_t)dst >> 32));
    const u64* ssrc = (const u64*)(((u64)shi << 32) | slo);
    u64* sdst = (u64*)(((u64)dhi << 32) | dlo);
    // let the streamers fill their queues
    for (int z = 0; z < 200; ++z) __builtin_amdgcn_s_sleep(64);
    u64 t0 = __builtin_amdgcn_s_memrealtime();
    unsigned fails = 0;
    for (int i = 1; i <= iters; ++i) {
        const unsigned tag = tag0 + i;
        const u64 val = ((u64)tag << 32) | 7u;
        auto put = [&]() {
            if (MODE == 0 || MODE == 1) { if (lane == 0) st<2>(dst, val); }
            else sst(sdst, val);
        };
        if (b == wa) put();
        unsigned spins = 0;
        for (;;) {
            unsigned got;
            if (MODE == 0 || MODE == 3) { const u64 x = ld<2>(src); got = __builtin_amdgcn_readfirstlane((unsigned)(x >> 32)); }
            else { const u64 x = sld(ssrc); got = (unsigned)(x >> 32); }
            if (got == tag) break;
            if (++spins > 200000u) { ++fails; break; }
        }
        if (b == wb) put();
        if (fails) break;
    }
    u64 t1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) { out[b == wa ? 0 : 2] = t1 - t0; out[b == wa ? 1 : 3] = fails; }
    __hip_atomic_store(&done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <int MODE> static void run_own(const char* nm, int wa, int wb, u64* ga, u64* gb, u64* out, unsigned& tag, const u32x4* W, size_t wn, int streamers) {
    const int iters = 200;
    CK(hipMemset(out, 0, 64));
    hipLaunchKernelGGL((pingpong_own<MODE>), dim3(256), dim3(320), 0, 0, ga, gb, wa, wb, iters, tag, out, W, wn, streamers);
    CK(hipDeviceSynchronize());
    tag += iters + 8;
    u64 h[4]; CK(hipMemcpy(h, out, 32, hipMemcpyDeviceToHost));
    if (h[1] || h[3]) printf("%-44s wg %3d <-> %3d  own stream of %d waves: NEVER ARRIVED\n", nm, wa, wb, streamers);
    else printf("%-44s wg %3d <-> %3d  own stream of %d waves: %6.2f us per hop\n", nm, wa, wb, streamers, (double)h[0] * 0.01 / iters / 2);
}

int main() {
    u64 *ga, *gb, *out; u32x4* W;
    const size_t wbytes = (size_t)2 << 30;
    CK(hipMalloc(&ga, 4096)); CK(hipMalloc(&gb, 4096)); CK(hipMalloc(&out, 4096)); CK(hipMalloc(&W, wbytes));
    CK(hipMemset(ga, 0, 4096)); CK(hipMemset(gb, 0, 4096)); CK(hipMemset(W, 1, wbytes));
    unsigned tag = 1;
    const size_t wn = wbytes / 16;
    for (int load : {0}) {
        for (int pair = 0; pair < 3; ++pair) {
            const int wa = 0, wb = pair == 0 ? 8 : pair == 1 ? 1 : 133;      // same XCD (0 and 8), neighbouring XCD, far
#define RUN(S, L, NM) run<S, L>(NM, wa, wb, ga, gb, out, tag, W, wn, load);
            RUN(2, 2, "store sc1      / load sc1      (agent: today)")
            RUN(3, 3, "store sc0 sc1  / load sc0 sc1  (system)")
            RUN(1, 1, "store sc0      / load sc0      (workgroup)")
            RUN(0, 1, "store plain    / load sc0")
            RUN(0, 2, "store plain    / load sc1")
            RUN(2, 1, "store sc1      / load sc0")
            RUN(1, 2, "store sc0      / load sc1")
            RUN(0, 0, "store plain    / load plain")
        }
    }
    for (int streamers : {0, 4}) for (int pair = 0; pair < 2; ++pair) {
        const int wa = 0, wb = pair == 0 ? 8 : 133;
        run_own<0>("vector store sc1 / vector load sc1", wa, wb, ga, gb, out, tag, W, wn, streamers);
        run_own<1>("vector store sc1 / SCALAR load glc", wa, wb, ga, gb, out, tag, W, wn, streamers);
        run_own<2>("SCALAR store glc / SCALAR load glc", wa, wb, ga, gb, out, tag, W, wn, streamers);
        run_own<3>("SCALAR store glc / vector load sc1", wa, wb, ga, gb, out, tag, W, wn, streamers);
    }
    return 0;
}
