// Native multi-GPU communicator of the C ABI: RCCL collectives on the caller's HIP stream, one
// process (or thread) per GPU.  librccl is opened at run time on first use, so single-GPU callers
// do not need it and a process that already holds a copy (PyTorch ships one) shares that copy.
#include <dlfcn.h>
#include <cstring>
#include <mutex>
#include <new>

#include <rccl/rccl.h>

#include "common.h"

using namespace vs;

namespace {

struct RcclApi {
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

RcclApi g_api;
std::mutex g_api_mutex;

int load_rccl()
{
    std::lock_guard<std::mutex> lock(g_api_mutex);
    if (g_api.ok) return VS_OK;
    void *lib = nullptr;
    const char *names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    for (const char *n : names) {
        lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (lib) break;
    }
    VS_CHECK(lib != nullptr, VS_ERR_UNSUPPORTED, "librccl.so.1 cannot be opened: %s", dlerror());
    g_api.GetUniqueId = (decltype(g_api.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    g_api.CommInitRank = (decltype(g_api.CommInitRank))dlsym(lib, "ncclCommInitRank");
    g_api.CommDestroy = (decltype(g_api.CommDestroy))dlsym(lib, "ncclCommDestroy");
    g_api.AllReduce = (decltype(g_api.AllReduce))dlsym(lib, "ncclAllReduce");
    g_api.AllGather = (decltype(g_api.AllGather))dlsym(lib, "ncclAllGather");
    g_api.GetErrorString = (decltype(g_api.GetErrorString))dlsym(lib, "ncclGetErrorString");
    VS_CHECK(g_api.GetUniqueId && g_api.CommInitRank && g_api.CommDestroy && g_api.AllReduce && g_api.AllGather && g_api.GetErrorString,
             VS_ERR_UNSUPPORTED, "librccl lacks a required symbol");
    g_api.ok = true;
    return VS_OK;
}

#define VS_RCCL(expr)                                                                           \
    do {                                                                                        \
        ncclResult_t r_ = (expr);                                                               \
        if (r_ != ncclSuccess) {                                                                \
            vs::set_error("%s failed: %s", #expr, g_api.GetErrorString(r_));                    \
            return VS_ERR_HIP;                                                                  \
        }                                                                                       \
    } while (0)

}  // namespace

struct vs_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

namespace vs {

int comm_rank(const vs_comm *c) { return c->rank; }
int comm_world(const vs_comm *c) { return c->world; }

int comm_allreduce_min_u64(vs_comm *c, unsigned long long *buf, int64_t n, hipStream_t s)
{
    VS_CHECK(c && c->comm, VS_ERR_STATE, "communicator not initialised");
    if (n <= 0) return VS_OK;
    VS_RCCL(g_api.AllReduce(buf, buf, (size_t)n, ncclUint64, ncclMin, c->comm, s));
    return VS_OK;
}

// every rank's n doubles at `send` end up at recv + rank * n on every rank
int comm_allgather(vs_comm *c, const double *send, double *recv, int64_t n, hipStream_t s)
{
    VS_CHECK(c && c->comm, VS_ERR_STATE, "communicator not initialised");
    if (n <= 0) return VS_OK;
    VS_RCCL(g_api.AllGather(send, recv, (size_t)n, ncclDouble, c->comm, s));
    return VS_OK;
}

int comm_allreduce(vs_comm *c, double *buf, int64_t n, bool take_max, hipStream_t s)
{
    VS_CHECK(c && c->comm, VS_ERR_STATE, "communicator not initialised");
    if (n <= 0) return VS_OK;
    VS_RCCL(g_api.AllReduce(buf, buf, (size_t)n, ncclDouble, take_max ? ncclMax : ncclSum, c->comm, s));
    return VS_OK;
}

}  // namespace vs

extern "C" {

int vs_comm_unique_id(void *id128)
{
    VS_CHECK(id128 != nullptr, VS_ERR_INVALID, "null id buffer");
    static_assert(sizeof(ncclUniqueId) == VS_COMM_ID_BYTES, "unique id size");
    VS_TRY(load_rccl());
    ncclUniqueId id;
    VS_RCCL(g_api.GetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return VS_OK;
}

int vs_comm_create(vs_comm_handle *out, int32_t rank, int32_t world, const void *id128)
{
    VS_CHECK(out != nullptr, VS_ERR_INVALID, "null handle pointer");
    *out = nullptr;
    VS_CHECK(world >= 1 && rank >= 0 && rank < world && id128, VS_ERR_INVALID, "bad rank %d of %d", rank, world);
    VS_TRY(load_rccl());
    vs_comm *c = new (std::nothrow) vs_comm();
    VS_CHECK(c != nullptr, VS_ERR_NOMEM, "out of host memory");
    c->rank = rank; c->world = world;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclResult_t r = g_api.CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, g_api.GetErrorString(r));
        delete c;
        return VS_ERR_HIP;
    }
    *out = c;
    return VS_OK;
}

int vs_comm_destroy(vs_comm_handle c)
{
    if (!c) return VS_OK;
    if (c->comm && g_api.ok) (void)g_api.CommDestroy(c->comm);
    delete c;
    return VS_OK;
}

int vs_comm_allreduce_sum(vs_comm_handle c, double *device_buf, int64_t n, void *hip_stream)
{
    VS_CHECK(c != nullptr && (n == 0 || device_buf), VS_ERR_INVALID, "null argument");
    return comm_allreduce(c, device_buf, n, false, (hipStream_t)hip_stream);
}

int vs_comm_allgather(vs_comm_handle c, const double *device_send, double *device_recv, int64_t n, void *hip_stream)
{
    VS_CHECK(c != nullptr && (n == 0 || (device_send && device_recv)), VS_ERR_INVALID, "null argument");
    return comm_allgather(c, device_send, device_recv, n, (hipStream_t)hip_stream);
}

}  // extern "C"
