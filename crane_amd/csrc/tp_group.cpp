#include "tp_group.h"

#include <atomic>
#include <cstdlib>
#include <cmath>
#include <set>

namespace cm {

TpGroup::TpGroup(int n_ranks, const int32_t* devices, int first_device, uint32_t collective) : n(n_ranks) {
    if (n < 2 || n > TP_MAX_RANKS) throw CmError(CM_ERR_INVALID, "in-process tensor parallelism: tp_size must be 2 .. 8");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) throw CmError(CM_ERR_DEVICE, "no HIP device");
    shared.n = n;
    std::set<int> distinct;
    for (int r = 0; r < n; ++r) {
        const int d = devices ? devices[r] : first_device + r;
        if (d < 0 || d >= ndev) throw CmError(CM_ERR_INVALID, "tp_devices: device ordinal out of range (" + std::to_string(d) + " of " + std::to_string(ndev) + ")");
        shared.devs.push_back(d);
        distinct.insert(d);
    }
    // every rank on its own GPU (production), or every rank on ONE GPU (test mode: proves sharding + exchange on a 1-GPU box)
    if (distinct.size() != (size_t)n && distinct.size() != 1)
        throw CmError(CM_ERR_INVALID, "tp_devices: either tp_size distinct devices or the same device tp_size times");
    shared.same_device = distinct.size() == 1;
    if (collective == CM_TP_COLL_RCCL && shared.same_device)
        throw CmError(CM_ERR_INVALID, "RCCL cannot place two ranks on one device: use the peer-store collective (tp_collective = 0 or CM_TP_COLL_PEER)");
    shared.use_peer = shared.same_device || collective == CM_TP_COLL_PEER;
    static std::atomic<uint32_t> g_groups{0};
    const uint32_t gi = g_groups.fetch_add(1);
    if (const char* e = getenv("CM_TP_EPOCH_BASE")) shared.epoch_base = (uint32_t)strtoul(e, nullptr, 0) * gi * 1000003u;
    if (shared.same_device) {
        // Ranks that share a device need one hardware queue per rank stream: with fewer, a rank's spinning exchange kernel sits
        // in the queue in front of the peer's kernel it waits for and every collective runs into its 2 s bound.  The HIP runtime
        // reads GPU_MAX_HW_QUEUES (default 4) once, when it starts -- the caller has to export it before the first HIP call.
        int hwq = 4;
        if (const char* e = getenv("GPU_MAX_HW_QUEUES")) hwq = atoi(e);
        if (n > hwq)
            throw CmError(CM_ERR_INVALID, "in-process tensor parallelism with every rank on ONE device (test mode) needs GPU_MAX_HW_QUEUES >= tp_size (" +
                          std::to_string(n) + ") exported before the HIP runtime starts; it is " + std::to_string(hwq));
    }
    if (!shared.use_peer) Rccl::unique_id(&shared.uid);
    errs.resize((size_t)n);
    for (int r = 1; r < n; ++r) peers.emplace_back(new Model());
    // the worker threads start LAST: nothing below can throw past joinable threads (std::terminate)
    try {
        for (int r = 1; r < n; ++r) th.emplace_back([this, r] { worker(r); });
    } catch (...) {
        { std::lock_guard<std::mutex> g(mu); quit = true; }
        cv_go.notify_all();
        for (auto& t : th) t.join();
        throw CmError(CM_ERR_DEVICE, "in-process tensor parallelism: cannot start the rank threads");
    }
}

TpGroup::~TpGroup() {
    {
        std::lock_guard<std::mutex> g(mu);
        quit = true;
    }
    cv_go.notify_all();
    for (auto& t : th) t.join();
    // the ranks' Models are destroyed on their own devices
    for (int r = n - 1; r >= 1; --r) { (void)hipSetDevice(shared.devs[(size_t)r]); peers[(size_t)r - 1].reset(); }
    (void)hipSetDevice(shared.devs[0]);
}

void TpGroup::run_rank(int r, const std::function<void(int)>& f) {
    try {
        (void)hipSetDevice(shared.devs[(size_t)r]);
        f(r);
        Model& m = model(r);
        if (m.rccl) m.rccl->check();
    } catch (...) {
        errs[(size_t)r] = std::current_exception();
        shared.fail();
        // RCCL transport: the other ranks may be blocked INSIDE a collective (a kernel waiting for this rank's contribution behind
        // a hipStreamSynchronize) -- no host-side rendezvous can release that.  ncclCommAbort of every rank's communicator does; it
        // may be called from this thread while the owner is blocked.  The group is dead afterwards (run()).
        if (!shared.use_peer && comms_ready.load() && !symmetric_error(errs[(size_t)r]))
            for (int q = 0; q < n; ++q) {
                Model* mq = q == 0 ? rank0 : peers[(size_t)q - 1].get();
                if (q != r && mq && mq->rccl) mq->rccl->abort_comm();
            }
    }
}

// errors every rank raises alike, before any exchange step: argument / range / state checks (the ranks hold identical state)
bool TpGroup::symmetric_error(const std::exception_ptr& e) {
    try { std::rethrow_exception(e); }
    catch (const CmError& x) { return x.code == CM_ERR_INVALID || x.code == CM_ERR_RANGE || x.code == CM_ERR_UNSUPPORTED; }
    catch (...) {}
    return false;
}

void TpGroup::worker(int r) {
    uint64_t seen = 0;
    for (;;) {
        const std::function<void(int)>* f = nullptr;
        {
            std::unique_lock<std::mutex> g(mu);
            cv_go.wait(g, [&] { return quit || gen != seen; });
            if (quit) return;
            seen = gen;
            f = job;
        }
        run_rank(r, *f);
        {
            std::lock_guard<std::mutex> g(mu);
            if (--pending == 0) cv_done.notify_all();
        }
    }
}

void TpGroup::run(const std::function<void(int)>& f) {
    if (dead) throw CmError(CM_ERR_DEVICE, "tensor-parallel group is dead after a rank failure (" + dead_why + "): destroy the handle and create a new one");
    {
        std::lock_guard<std::mutex> g(mu);
        for (auto& e : errs) e = nullptr;
        { std::lock_guard<std::mutex> g2(shared.mu); shared.failed = false; shared.arrived = 0; }
        cb.verdict.clear();
        job = &f;
        pending = n - 1;
        ++gen;
    }
    cv_go.notify_all();
    run_rank(0, f);
    {
        std::unique_lock<std::mutex> g(mu);
        cv_done.wait(g, [&] { return pending == 0; });
        job = nullptr;
    }
    (void)hipSetDevice(shared.devs[0]);
    // report the ROOT cause: a rank that failed on its own, not the ranks that were released from a wait because of it
    std::exception_ptr first = nullptr, released = nullptr;
    int n_err = 0, n_sym = 0;
    for (auto& e : errs) if (e) { ++n_err; n_sym += symmetric_error(e) ? 1 : 0; }
    if (n_err > 0 && !(n_err == n && n_sym == n)) {
        dead = true;
        for (auto& e : errs) {
            if (!e || !dead_why.empty()) continue;
            try { std::rethrow_exception(e); } catch (const std::exception& x) { if (std::string(x.what()).find("another rank failed") == std::string::npos) dead_why = x.what(); } catch (...) { dead_why = "unknown exception"; }
        }
        if (dead_why.empty()) dead_why = "a rank failed";
        if (dead_why.size() > 160) dead_why.resize(160);
    }
    for (auto& e : errs) {
        if (!e) continue;
        bool rel = false;
        try { std::rethrow_exception(e); }
        catch (const CmError& x) { rel = std::string(x.what()).find("another rank failed") != std::string::npos; }
        catch (...) {}
        if (rel) { if (!released) released = e; } else if (!first) first = e;
    }
    if (first) std::rethrow_exception(first);
    if (released) std::rethrow_exception(released);
    comms_ready.store(true);
}

}  // namespace cm

// ---- self-test of the peer-store collectives alone (no model): n rank threads on one device, random vectors, the sums and the
// gathers checked on the host.  cm_debug_peer_selftest (test hook; tests/test_gpu_tp_group.py)
namespace cm {
long peer_selftest(int n, int device, int iters, int count) {
    if (n < 2 || n > TP_MAX_RANKS || count < 1 || iters == 0) throw CmError(CM_ERR_INVALID, "peer_selftest arguments");
    PeerShared ps;
    if (iters < 0) { iters = -iters; ps.epoch_base = 0xFFFFFFFDu; }      // walk the epoch counter across its 32-bit wrap (kernels_tp.hip)
    ps.n = n; ps.devs.assign((size_t)n, device); ps.same_device = true; ps.use_peer = true;
    std::vector<long> bad((size_t)n, 0);
    std::vector<std::vector<float>> in((size_t)n);
    std::vector<std::exception_ptr> errs((size_t)n);
    for (int r = 0; r < n; ++r) {
        in[(size_t)r].resize((size_t)count);
        for (int i = 0; i < count; ++i) in[(size_t)r][(size_t)i] = (float)((int)(fmix32((uint32_t)(r * 7919 + i)) % 2001) - 1000) / 64.f;
    }
    auto body = [&](int r) {
        try {
            (void)hipSetDevice(device);
            hipStream_t s;
            CM_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
            Rccl cc;
            cc.init_peer(&ps, r, 256, s);
            float *d_in, *d_out, *d_g;
            CM_HIP(hipMalloc((void**)&d_in, (size_t)count * 4));
            CM_HIP(hipMalloc((void**)&d_out, (size_t)count * 4));
            CM_HIP(hipMalloc((void**)&d_g, (size_t)count * 4 * n));
            std::vector<float> out((size_t)count), gat((size_t)count * n), mine((size_t)count);
            for (int it = 0; it < iters; ++it) {
                for (int i = 0; i < count; ++i) mine[(size_t)i] = in[(size_t)r][(size_t)i] + (float)it;
                CM_HIP(hipMemcpyAsync(d_in, mine.data(), (size_t)count * 4, hipMemcpyHostToDevice, s));
                cc.all_reduce_sum_f32(d_in, d_out, (size_t)count, s);
                CM_HIP(hipMemcpyAsync(d_g + (size_t)r * count, d_in, (size_t)count * 4, hipMemcpyDeviceToDevice, s));
                cc.all_gather(d_g + (size_t)r * count, d_g, (size_t)count * 4, s);
                CM_HIP(hipMemcpyAsync(out.data(), d_out, (size_t)count * 4, hipMemcpyDeviceToHost, s));
                CM_HIP(hipMemcpyAsync(gat.data(), d_g, (size_t)count * 4 * n, hipMemcpyDeviceToHost, s));
                CM_HIP(hipStreamSynchronize(s));
                cc.check();
                for (int i = 0; i < count; ++i) {
                    float want = 0.f;
                    for (int q = 0; q < n; ++q) { const float v = in[(size_t)q][(size_t)i] + (float)it; want = q == 0 ? v : want + v; }
                    if (out[(size_t)i] != want) ++bad[(size_t)r];
                    for (int q = 0; q < n; ++q) if (gat[(size_t)q * count + i] != in[(size_t)q][(size_t)i] + (float)it) ++bad[(size_t)r];
                }
            }
            (void)hipFree(d_in); (void)hipFree(d_out); (void)hipFree(d_g);
            ps.arrive_and_wait();           // nobody frees an inbox a peer may still write
            (void)hipStreamDestroy(s);
        } catch (...) { errs[(size_t)r] = std::current_exception(); ps.fail(); }
    };
    std::vector<std::thread> th;
    for (int r = 1; r < n; ++r) th.emplace_back(body, r);
    body(0);
    for (auto& t : th) t.join();
    for (auto& e : errs) if (e) std::rethrow_exception(e);
    long tot = 0;
    for (long b : bad) tot += b;
    return tot;
}
}  // namespace cm
