// Tensor-parallel collectives.  Two transports behind one interface:
//  * RCCL over xGMI (the default whenever every rank has its own GPU), resolved with dlopen so that a TP=1 process never
//    touches RCCL and a TP>1 process shares whatever librccl the host process already mapped (e.g. the one torch ships)
//    instead of loading a second copy.  One communicator rank per Model: one process per GPU (SPMD, cm_opts.tp_unique_id
//    from the caller), or -- cm_opts.tp_mode = CM_TP_IN_PROCESS -- one library thread per GPU inside ONE handle.
//  * peer-store collectives (kernels_tp.hip) for in-process groups: one-shot push all-reduce / all-gather over peer-visible
//    memory; the only transport when several ranks share ONE device (the single-GPU test mode -- RCCL refuses two ranks on a
//    device), opt-in otherwise (cm_opts.tp_collective = CM_TP_COLL_PEER).
//
// The reference has no distributed code at all ("Multi-GPU tensor
// parallelism is not yet supported", crane-serve/README.md:624); every call site here is
// new design: one all-reduce of the [H] f32 partial residual after o_proj and after
// down_proj (row-parallel matmuls), one all-gather of arg-max partials / logits after the
// vocab-sharded lm_head.
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <condition_variable>
#include <mutex>
#include <vector>

#include "kernels.h"

namespace cm {

struct UniqueId { char internal[128]; };   // == ncclUniqueId (rccl.h:40-43)

// State shared by the ranks of ONE in-process group (owned by the TpGroup, tp_group.h): the rendezvous the rank threads use
// while they set themselves up, the abort flag that releases every wait when a rank fails, and the peer-store inboxes.
struct PeerShared {
    int n = 1;
    std::vector<int> devs;                 // device ordinal of rank r
    bool same_device = false;              // every rank on ONE device (test mode)
    bool use_peer = false;                 // transport: peer-store kernels (else RCCL, one communicator rank per thread)
    UniqueId uid{};                        // RCCL id of the group (created by the group before the ranks start)
    size_t cap = (size_t)1 << 20;          // granules per (parity, source) inbox slot: larger collectives run in pieces
    uint32_t epoch_base = 0;               // first epoch - 1 of this group
    unsigned long long* inbox[TP_MAX_RANKS] = {};
    // abortable barrier
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t phase = 0;
    bool failed = false;
    void fail();                            // a rank threw: release every waiter (they throw CM_ERR_DEVICE)
    void arrive_and_wait();                 // all n ranks; throws if the group failed meanwhile
};

struct Rccl {
    void* lib = nullptr;
    std::atomic<void*> comm{nullptr};       // taken with exchange(nullptr) by whoever destroys / aborts it: a failing peer's thread may race the owner (tp_group.cpp)
    int nranks = 1, rank = 0;
    bool fake = false;     // cm_opts.debug_flags & CM_DEBUG_TP_LOCAL: no communicator; all-reduce = local copy, all-gather = no-op, so one
                           // process can run ONE rank's shard and be compared with the oracle on the same shard
    // peer-store transport (in-process group)
    PeerShared* peer = nullptr;
    uint32_t* ctl = nullptr;               // device: epoch, finish ticket
    uint32_t* h_err = nullptr;             // pinned + mapped: raised by a collective whose wait timed out
    uint32_t* d_err = nullptr;             // the same word as the device addresses it
    bool inbox_uncached = false;           // this rank's inbox is an uncached allocation (parked at destruction, never freed)
    size_t inbox_bytes = 0;
    int peer_blocks = 64;
    long max_spin = 2000000;               // ~2 s (a polled load + s_sleep is ~1 us)
    // resolved entry points
    int (*p_get_unique_id)(void*) = nullptr;
    int (*p_comm_init_rank)(void**, int, /*ncclUniqueId by value*/ UniqueId, int) = nullptr;
    int (*p_comm_destroy)(void*) = nullptr;
    int (*p_comm_abort)(void*) = nullptr;   // optional symbol (ncclCommAbort): releases the group's ranks when one of them fails
    int (*p_all_reduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*p_all_gather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    const char* (*p_get_error_string)(int) = nullptr;

    ~Rccl();
    void load();
    void init(int nranks, int rank, const void* unique_id128, hipStream_t s, bool local_only);
    void init_peer(PeerShared* ps, int rank, int num_cu, hipStream_t s);    // collective: every rank of the group calls it
    void all_reduce_sum_f32(const float* send, float* recv, size_t count, hipStream_t s);
    // gathers `bytes_per_rank` from every rank into recv (rank-major); send may alias its slot
    void all_gather(const void* send, void* recv, size_t bytes_per_rank, hipStream_t s);
    void abort_comm();                      // RCCL transport: ncclCommAbort (callable from another thread while this rank is blocked in a collective)
    void check();                           // after a host sync: throws CM_ERR_DEVICE if a peer-store wait timed out
    static void unique_id(void* out128);
};

}  // namespace cm
