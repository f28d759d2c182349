// Tensor-parallel collectives: RCCL over xGMI, resolved with dlopen so that a TP=1
// process never touches RCCL and a TP>1 process shares whatever librccl the host
// process already mapped (e.g. the one torch ships) instead of loading a second copy.
//
// The reference has no distributed code at all ("Multi-GPU tensor
// parallelism is not yet supported", crane-serve/README.md:624); every call site here is
// new design: one all-reduce of the [H] f32 partial residual after o_proj and after
// down_proj (row-parallel matmuls), one all-gather of arg-max partials / logits after the
// vocab-sharded lm_head.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace cm {

struct UniqueId { char internal[128]; };   // == ncclUniqueId (rccl.h:40-43)

struct Rccl {
    void* lib = nullptr;
    void* comm = nullptr;
    int nranks = 1, rank = 0;
    bool fake = false;     // cm_opts.debug_flags & CM_DEBUG_TP_LOCAL: no communicator; all-reduce = local copy, all-gather = no-op, so one
                           // process can run ONE rank's shard and be compared with the oracle on the same shard
    // resolved entry points
    int (*p_get_unique_id)(void*) = nullptr;
    int (*p_comm_init_rank)(void**, int, /*ncclUniqueId by value*/ UniqueId, int) = nullptr;
    int (*p_comm_destroy)(void*) = nullptr;
    int (*p_all_reduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*p_all_gather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    const char* (*p_get_error_string)(int) = nullptr;

    ~Rccl();
    void load();
    void init(int nranks, int rank, const void* unique_id128, hipStream_t s, bool local_only);
    void all_reduce_sum_f32(const float* send, float* recv, size_t count, hipStream_t s);
    // gathers `bytes_per_rank` from every rank into recv (rank-major); send may alias its slot
    void all_gather(const void* send, void* recv, size_t bytes_per_rank, hipStream_t s);
    static void unique_id(void* out128);
};

}  // namespace cm
