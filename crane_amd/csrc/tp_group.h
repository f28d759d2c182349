// In-process tensor-parallel group: ONE cm_model handle that owns all tp_size ranks (cm_opts.tp_mode = CM_TP_IN_PROCESS).
//
// Why: the reference hosts a model as ONE ModelBackend object moved into ONE engine thread of ONE process
// (crane-serve/src/lib.rs:1129-1132, engine/backend.rs:30); the SPMD form of tensor parallelism (one process per GPU, every
// process issuing the same cm_* calls) cannot be hosted there.  Here the library does the fan-out: rank r is a full Model
// (its shard of the weights, its KV heads, its stream and hipGraphs) on device tp_devices[r]; every cm_* call on the handle
// runs on all ranks at once -- rank 0 on the calling thread, ranks 1..n-1 on library worker threads that live as long as the
// handle -- and returns rank 0's results (the ranks hold identical sequences, page tables and, after each exchange step,
// bit-identical activations, so any rank's result is the result).  The exchange steps are RCCL (one communicator rank per
// thread) or the peer-store collectives of kernels_tp.hip (tp.h).
#pragma once
#include <condition_variable>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <atomic>

#include "model.h"
#include "tp.h"

namespace cm {

struct TpGroup {
    int n = 1;
    PeerShared shared;
    Model* rank0 = nullptr;                          // the handle's own Model
    std::vector<std::unique_ptr<Model>> peers;       // ranks 1 .. n-1
    Model& model(int r) { return r == 0 ? *rank0 : *peers[(size_t)r - 1]; }

    TpGroup(int n_ranks, const int32_t* devices, int first_device, uint32_t collective);
    ~TpGroup();
    // A group whose ranks failed ASYMMETRICALLY (one threw, the others did not -- or were released from a wait / an RCCL call
    // because of it) is out of step for good: sequences, page tables and the collectives' epoch counters no longer agree.  The
    // handle is then DEAD: every later call returns CM_ERR_DEVICE at once (the caller destroys the handle and creates a new
    // one); errors every rank raises alike (an invalid argument, a range check -- thrown before any exchange) do not kill it.
    bool dead = false;
    std::string dead_why;
    // set once a run() completed without error: every rank's Model::rccl is constructed and stays put.  Until then (the run that builds
    // the ranks: alloc_runtime) a failing rank must not touch its peers' half-built communicators (run_rank)
    std::atomic<bool> comms_ready{false};
    // f(rank) on every rank concurrently; returns when all are done.  A rank that throws aborts the group's rendezvous
    // (PeerShared::fail) so that no other rank waits for it forever; the exception of the lowest failing rank is rethrown.
    void run(const std::function<void(int)>& f);

    // generate(): the token callback runs on rank 0 only; the other ranks wait for its verdict (continue / stop) token by token
    struct CbSync { std::vector<int> verdict; };
    CbSync cb;

private:
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    uint64_t gen = 0;
    int pending = 0;
    bool quit = false;
    const std::function<void(int)>* job = nullptr;
    std::vector<std::exception_ptr> errs;
    void worker(int r);
    void run_rank(int r, const std::function<void(int)>& f);
    static bool symmetric_error(const std::exception_ptr& e);
};

}  // namespace cm
