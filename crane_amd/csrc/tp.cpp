#include "tp.h"

#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
#include <string>

#include "model.h"

namespace cm {

// RCCL (= NCCL API) enums we need: ncclFloat32 = 7, ncclChar/ncclInt8 = 0, ncclSum = 0
static constexpr int kNcclFloat32 = 7, kNcclInt8 = 0, kNcclSum = 0;

static void* open_rccl() {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so", nullptr};
    for (int i = 0; names[i]; ++i) {
        if (void* h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL)) return h;
    }
    return nullptr;
}

void Rccl::load() {
    if (lib) return;
    lib = open_rccl();
    if (!lib) throw CmError(CM_ERR_DEVICE, std::string("cannot dlopen librccl: ") + (dlerror() ? dlerror() : "?"));
    auto sym = [&](const char* n) {
        void* p = dlsym(lib, n);
        if (!p) throw CmError(CM_ERR_DEVICE, std::string("librccl lacks symbol ") + n);
        return p;
    };
    p_get_unique_id = (decltype(p_get_unique_id))sym("ncclGetUniqueId");
    p_comm_init_rank = (decltype(p_comm_init_rank))sym("ncclCommInitRank");
    p_comm_destroy = (decltype(p_comm_destroy))sym("ncclCommDestroy");
    p_all_reduce = (decltype(p_all_reduce))sym("ncclAllReduce");
    p_all_gather = (decltype(p_all_gather))sym("ncclAllGather");
    p_get_error_string = (decltype(p_get_error_string))sym("ncclGetErrorString");
    p_comm_abort = (decltype(p_comm_abort))dlsym(lib, "ncclCommAbort");          // (optional)
}

void Rccl::abort_comm() {
    if (!p_comm_abort) return;
    void* c = comm.exchange(nullptr);           // exactly one caller gets the communicator (two ranks failing at once both call this on every peer)
    if (c) (void)p_comm_abort(c);
}

#define CM_NCCL(expr)                                                                           \
    do {                                                                                        \
        int _r = (expr);                                                                        \
        if (_r != 0)                                                                            \
            throw CmError(CM_ERR_DEVICE, std::string(#expr) + ": " +                             \
                                             (p_get_error_string ? p_get_error_string(_r) : "rccl error")); \
    } while (0)

// Uncached inbox blocks are never handed back to the allocator (see init_peer): a destroyed group parks them here and the
// next group on that device takes them again.
namespace {
struct ParkedInbox { int dev; size_t bytes; void* p; };
std::mutex g_park_mu;
std::vector<ParkedInbox> g_parked;
void* take_parked(int dev, size_t bytes) {
    std::lock_guard<std::mutex> g(g_park_mu);
    for (size_t i = 0; i < g_parked.size(); ++i)
        if (g_parked[i].dev == dev && g_parked[i].bytes == bytes) { void* p = g_parked[i].p; g_parked.erase(g_parked.begin() + (long)i); return p; }
    return nullptr;
}
}  // namespace

Rccl::~Rccl() {
    if (void* c = comm.exchange(nullptr)) { if (p_comm_destroy) (void)p_comm_destroy(c); }
    // the library stays mapped for the life of the process
    if (peer) {
        if (peer->inbox[rank]) {
            if (inbox_uncached) { std::lock_guard<std::mutex> g(g_park_mu); g_parked.push_back({peer->devs[(size_t)rank], inbox_bytes, peer->inbox[rank]}); }
            else (void)hipFree(peer->inbox[rank]);
            peer->inbox[rank] = nullptr;
        }
        if (ctl) (void)hipFree(ctl);
        if (h_err) (void)hipHostFree(h_err);
    }
}

// ---- in-process group: rendezvous of the rank threads + peer-store transport ----------------------------------------

void PeerShared::fail() {
    std::lock_guard<std::mutex> g(mu);
    failed = true;
    cv.notify_all();
}

void PeerShared::arrive_and_wait() {
    std::unique_lock<std::mutex> g(mu);
    if (failed) throw CmError(CM_ERR_DEVICE, "tensor-parallel group: another rank failed");
    const uint64_t my = phase;
    if (++arrived == n) { arrived = 0; ++phase; cv.notify_all(); return; }
    cv.wait(g, [&] { return phase != my || failed; });
    if (phase == my) throw CmError(CM_ERR_DEVICE, "tensor-parallel group: another rank failed");
}

void Rccl::init_peer(PeerShared* ps, int r, int num_cu, hipStream_t s) {
    peer = ps; nranks = ps->n; rank = r;
    // inbox: [2 parities][n source ranks][cap] granules on THIS rank's device, written by every rank.  Uncached (fine-grained)
    // where the runtime offers it: remote stores must be visible to this device's loads without a cache maintenance step
    const size_t bytes = (size_t)2 * ps->n * ps->cap * sizeof(unsigned long long);
    // Ranks on ONE device (test mode): ordinary device memory -- the granules are written and polled with system-scope
    // (sc0 sc1) accesses, which is all the persistent decode kernel's cross-XCD hand-offs need as well.  (Seen on hardware,
    // ROCm 7.2: a range that had been hipExtMallocWithFlags(Uncached), freed, and handed out again by a plain hipMalloc -- it
    // became a later handle's split-K workspace -- returned stale data between two kernels of one stream until the whole range
    // had been rewritten once; uncached allocations are therefore made only where they are needed, never in the test mode.)
    void* p = nullptr;
    hipError_t e = hipErrorUnknown;
    inbox_bytes = bytes;
    if (!ps->same_device) {
        // real peers: remote stores must be visible to this device's loads without a cache maintenance step -> uncached
        // (fine-grained) memory; parked, not freed, when the group goes (the recycling hazard above)
        p = take_parked(ps->devs[(size_t)r], bytes);
        e = p ? hipSuccess : hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached);
        if (e != hipSuccess) { (void)hipGetLastError(); e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained); }
        if (e != hipSuccess) {
            // Coarse-grained hipMalloc memory is NOT an option between devices: a peer's posted stores polled with relaxed
            // system-scope loads are not guaranteed to become visible there (stale granules, spurious 2 s time-outs).  Refuse;
            // the caller can take RCCL (cm_opts.tp_collective = CM_TP_COLL_RCCL, the default on distinct devices).
            (void)hipGetLastError();
            throw CmError(CM_ERR_UNSUPPORTED, "peer-store collective: no uncached / fine-grained device memory for the inbox on device " +
                          std::to_string(ps->devs[(size_t)r]) + "; use tp_collective = CM_TP_COLL_RCCL");
        }
        inbox_uncached = true;
    } else {
        e = hipMalloc(&p, bytes);           // ranks on ONE device: ordinary memory (see above)
    }
    if (e != hipSuccess) throw CmError(CM_ERR_OOM, "peer inbox allocation failed");
    if (hipMemsetAsync(p, 0, bytes, s) != hipSuccess || hipMalloc((void**)&ctl, 4096) != hipSuccess ||
        hipMemsetAsync(ctl, 0, 4096, s) != hipSuccess ||
        hipHostMalloc((void**)&h_err, 64, hipHostMallocMapped) != hipSuccess)
        throw CmError(CM_ERR_OOM, "peer collective state allocation failed");
    *h_err = 0;
    d_err = h_err;
    { void* dp = nullptr; if (hipHostGetDevicePointer(&dp, h_err, 0) == hipSuccess && dp) d_err = (uint32_t*)dp; else (void)hipGetLastError(); }
    if (hipStreamSynchronize(s) != hipSuccess) throw CmError(CM_ERR_DEVICE, "peer inbox initialisation failed");
    if (getenv("CM_TP_DEBUG")) {
        std::vector<unsigned long long> hbuf(bytes / 8);
        (void)hipMemcpy(hbuf.data(), p, bytes, hipMemcpyDeviceToHost);
        size_t nz = 0; for (auto v : hbuf) nz += v != 0;
        fprintf(stderr, "[cm tp] rank %d inbox %p (%zu MiB): %zu non-zero granules after the clear; epoch base %u\n", r, p, bytes >> 20, nz, ps->epoch_base);
    }
    if (hipMemcpy(ctl, &ps->epoch_base, 4, hipMemcpyHostToDevice) != hipSuccess) throw CmError(CM_ERR_DEVICE, "peer epoch initialisation failed");
    ps->inbox[r] = (unsigned long long*)p;
    ps->arrive_and_wait();                                  // every inbox exists and is zeroed
    if (!ps->same_device)
        for (int d = 0; d < ps->n; ++d) {
            if (ps->devs[(size_t)d] == ps->devs[(size_t)r]) continue;
            int can = 0;
            (void)hipDeviceCanAccessPeer(&can, ps->devs[(size_t)r], ps->devs[(size_t)d]);
            if (!can) throw CmError(CM_ERR_DEVICE, "tensor-parallel group: no peer access between the devices");
            const hipError_t pe = hipDeviceEnablePeerAccess(ps->devs[(size_t)d], 0);
            if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) throw CmError(CM_ERR_DEVICE, "hipDeviceEnablePeerAccess failed");
            (void)hipGetLastError();
        }
    // spinning workgroups must never crowd out the peers' kernels when the ranks share one device (test mode): n ranks x
    // peer_blocks x 4 waves stays far below the device's wave slots
    peer_blocks = std::max(8, std::min(64, 2 * num_cu / std::max(1, ps->n) / 4));
    ps->arrive_and_wait();
}

void Rccl::check() {
    if (h_err && *h_err != 0) {
        *h_err = 0;
        if (ctl) { const uint32_t z = 0; (void)hipMemcpy(ctl + 2, &z, 4, hipMemcpyHostToDevice); }    // the next call waits again
        throw CmError(CM_ERR_DEVICE, "tensor-parallel exchange timed out: a rank of the group did not contribute (peer-store collective)");
    }
}

void Rccl::unique_id(void* out128) {
    Rccl r;
    r.load();
    UniqueId id;
    memset(&id, 0, sizeof id);
    if (r.p_get_unique_id(&id) != 0) throw CmError(CM_ERR_DEVICE, "ncclGetUniqueId failed");
    memcpy(out128, &id, sizeof id);
}

void Rccl::init(int n, int r, const void* unique_id128, hipStream_t, bool local_only) {
    if (local_only) { fake = true; nranks = n; rank = r; return; }
    if (!unique_id128) throw CmError(CM_ERR_INVALID, "tp_size > 1 needs cm_opts.tp_unique_id");
    load();
    nranks = n; rank = r;
    UniqueId id;
    memcpy(&id, unique_id128, sizeof id);
    void* c = nullptr;
    CM_NCCL(p_comm_init_rank(&c, n, id, r));
    comm.store(c);
}

static void peer_run(Rccl& r, int mode, const uint32_t* send, uint32_t* recv, size_t count, hipStream_t s) {
    PeerShared* ps = r.peer;
    for (size_t off = 0; off < count; off += ps->cap) {
        PeerCollArgs a{};
        for (int d = 0; d < ps->n; ++d) a.inbox[d] = ps->inbox[d];
        a.send = send + off; a.recv = recv + off; a.ctl = r.ctl; a.err = r.d_err; a.max_spin = r.max_spin;
        a.n = ps->n; a.me = r.rank; a.count = (int)std::min(ps->cap, count - off); a.recv_stride = (int)count; a.cap = ps->cap;
        const int blocks = std::max(1, std::min(r.peer_blocks, (a.count + 255) / 256));
        launch_peer_coll(mode, a, blocks, s);
    }
}

void Rccl::all_reduce_sum_f32(const float* send, float* recv, size_t count, hipStream_t s) {
    if (peer) { peer_run(*this, 0, (const uint32_t*)send, (uint32_t*)recv, count, s); return; }
    if (fake) {
        if (send != recv) (void)hipMemcpyAsync(recv, send, count * sizeof(float), hipMemcpyDeviceToDevice, s);
        return;
    }
    void* c = comm.load();
    if (!c) throw CmError(CM_ERR_DEVICE, "RCCL communicator aborted (a peer rank failed)");
    CM_NCCL(p_all_reduce(send, recv, count, kNcclFloat32, kNcclSum, c, s));
}

void Rccl::all_gather(const void* send, void* recv, size_t bytes_per_rank, hipStream_t s) {
    if (peer) {
        if (bytes_per_rank % 4) throw CmError(CM_ERR_INVALID, "peer all-gather needs 4-byte multiples");
        peer_run(*this, 1, (const uint32_t*)send, (uint32_t*)recv, bytes_per_rank / 4, s);
        return;
    }
    if (fake) return;
    void* c = comm.load();
    if (!c) throw CmError(CM_ERR_DEVICE, "RCCL communicator aborted (a peer rank failed)");
    CM_NCCL(p_all_gather(send, recv, bytes_per_rank, kNcclInt8, c, s));
}

}  // namespace cm
