#include "tp.h"

#include <dlfcn.h>

#include <cstdlib>
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
}

#define CM_NCCL(expr)                                                                           \
    do {                                                                                        \
        int _r = (expr);                                                                        \
        if (_r != 0)                                                                            \
            throw CmError(CM_ERR_DEVICE, std::string(#expr) + ": " +                             \
                                             (p_get_error_string ? p_get_error_string(_r) : "rccl error")); \
    } while (0)

Rccl::~Rccl() {
    if (comm && p_comm_destroy) (void)p_comm_destroy(comm);
    // the library stays mapped for the life of the process
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
    CM_NCCL(p_comm_init_rank(&comm, n, id, r));
}

void Rccl::all_reduce_sum_f32(const float* send, float* recv, size_t count, hipStream_t s) {
    if (fake) {
        if (send != recv) (void)hipMemcpyAsync(recv, send, count * sizeof(float), hipMemcpyDeviceToDevice, s);
        return;
    }
    CM_NCCL(p_all_reduce(send, recv, count, kNcclFloat32, kNcclSum, comm, s));
}

void Rccl::all_gather(const void* send, void* recv, size_t bytes_per_rank, hipStream_t s) {
    if (fake) return;
    CM_NCCL(p_all_gather(send, recv, bytes_per_rank, kNcclInt8, comm, s));
}

}  // namespace cm
