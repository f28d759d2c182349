// Exchange steps of an IN-PROCESS tensor-parallel group (cm_opts.tp_mode = CM_TP_IN_PROCESS with the peer-store collective):
// one-shot all-reduce / all-gather over peer-visible device memory -- every rank PUSHES its contribution into every rank's
// inbox (posted writes over xGMI, no read round trip) and then sums / copies what arrived in its OWN inbox, in rank order,
// so every rank ends with bit-identical results.  The reference has no multi-GPU code (crane-serve/README.md:622-624); the
// exchange points are DESIGN 6's: the [H] partial sums behind o_proj / out_proj and down_proj, the arg-max partials (or the
// logits shards) behind the vocabulary-sharded lm_head.
//
// Protocol (the persistent decode kernel's "the data is its own flag", kernels_engine.hip, stretched across devices): an
// element travels as ONE 8-byte {value, epoch} granule written with a system-scope store; the reader spins on the granule
// until its epoch is the current one.  No fences, no separate flags, no host involvement -- the launches are ordinary kernels
// and are captured into the decode hipGraph like any other.  Each rank counts its own collectives in device memory
// (ctl[0]); all ranks issue the same sequence, so the counters agree without being exchanged.  Inboxes are double-buffered
// by epoch parity: a rank can only be two collectives ahead of a peer after that peer contributed to the one in between,
// i.e. after it consumed the older buffer.  Why not fused into the consuming GEMV's prologue (SURVEY 8e's first idea): every
// one of the GEMV's ~512 workgroups needs the whole [H] vector, so each would read n x H granules (256 KB at TP = 8) of
// uncached memory -- 128 MB per projection against the 25 MB weight shard it streams; a 16-workgroup launch in between costs
// ~2.5 us inside a hipGraph.
//
// Wait bound: a peer that never arrives (a rank died, or -- all ranks on ONE device, the test mode -- a kernel that cannot be
// co-scheduled) ends the spin after ~2 s, raises the rank's host-visible error word and lets the launch finish with garbage;
// the host side turns that into CM_ERR_DEVICE at the next synchronisation point.
#include "kernels.h"

namespace cm {

__device__ __forceinline__ void st_sys64(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned long long ld_sys64(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// MODE 0: recv[i] = sum over ranks (rank order) of send[i] as f32.  MODE 1: recv[src * recv_stride + i] = send_src[i] (32-bit words).
template <int MODE>
__global__ void __launch_bounds__(256) peer_coll_kernel(PeerCollArgs a) {
    // ctl words are read and written with agent-scope accesses only (write-through stores, cache-bypassing loads) and live on
    // their own 4-KB block: a plain load could be served from an XCD's L2 line that predates the previous launch's update
    // (seen on hardware: two ranks' 64-byte control blocks sharing one 128-byte line, ranks on one device -- a rank then
    // re-used the previous epoch and consumed the previous collective's granules without waiting)
    const uint32_t e0 = __hip_atomic_load(&a.ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // epoch of THIS collective.  0 is "never written"; the inbox buffer is chosen by the epoch's parity, so consecutive epochs must
    // alternate in parity ALSO at the 32-bit wrap: 0xFFFFFFFF (odd) is followed by 2 (even) -- 0 and 1 are skipped.  (1 would put
    // two consecutive collectives into the same buffer, where a rank one collective ahead overwrites granules a slower peer has
    // not consumed yet; cm_debug_peer_selftest with iters < 0 starts the counter at 0xFFFFFFFD and walks across the wrap.)
    const uint32_t e = e0 + 1u == 0u ? 2u : e0 + 1u;
    const size_t par = (size_t)(e & 1u) * (size_t)a.n * a.cap;
    const int stride = (int)(gridDim.x * blockDim.x);
    const int i0 = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    // (1) push: my elements into slot `me` of every rank's inbox
    for (int i = i0; i < a.count; i += stride) {
        const unsigned long long g = ((unsigned long long)e << 32) | (unsigned long long)a.send[i];
        const size_t off = par + (size_t)a.me * a.cap + (size_t)i;
        for (int d = 0; d < a.n; ++d) st_sys64(a.inbox[d] + off, g);
    }
    // (2) collect: what the n ranks pushed into MY inbox.  After a first time-out the rank stops waiting for good (ctl[2]): a
    // group whose kernels cannot run side by side would otherwise spend the bound on every later collective
    const bool dead = __hip_atomic_load(&a.ctl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    bool timed_out = false;
    for (int i = i0; i < a.count; i += stride) {
        float acc = 0.f;
        for (int src = 0; src < a.n; ++src) {
            const unsigned long long* p = a.inbox[a.me] + par + (size_t)src * a.cap + (size_t)i;
            unsigned long long g = ld_sys64(p);
            for (long spin = 0; (uint32_t)(g >> 32) != e && spin < a.max_spin && !dead; ++spin) {
                __builtin_amdgcn_s_sleep(4);
                g = ld_sys64(p);
            }
            if ((uint32_t)(g >> 32) != e) timed_out = true;
            if (MODE == 0) { const float v = __uint_as_float((uint32_t)g); acc = src == 0 ? v : acc + v; }
            else a.recv[(size_t)src * a.recv_stride + i] = (uint32_t)g;
        }
        if (MODE == 0) a.recv[i] = __float_as_uint(acc);
    }
    if (timed_out) {
        __hip_atomic_fetch_or(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&a.ctl[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // (3) the last workgroup to finish advances this rank's epoch (every workgroup has read ctl[0] by then)
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = atomicAdd(&a.ctl[1], 1u);
        if (t == gridDim.x - 1) {
            __hip_atomic_store(&a.ctl[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&a.ctl[0], e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

void launch_peer_coll(int mode, const PeerCollArgs& a, int blocks, hipStream_t s) {
    if (mode == 0) hipLaunchKernelGGL(peer_coll_kernel<0>, dim3(blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(peer_coll_kernel<1>, dim3(blocks), dim3(256), 0, s, a);
}

}  // namespace cm
