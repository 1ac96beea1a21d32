// Shared helpers for the HIP translation units of libvslam_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include "vslam_hip.h"

namespace vs {

void set_error(const char *fmt, ...);

#define VS_HIP(expr)                                                                            \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            vs::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,      \
                          __LINE__);                                                            \
            return VS_ERR_HIP;                                                                  \
        }                                                                                       \
    } while (0)

#define VS_CHECK(cond, code, ...)                                                               \
    do {                                                                                        \
        if (!(cond)) {                                                                          \
            vs::set_error(__VA_ARGS__);                                                         \
            return (code);                                                                      \
        }                                                                                       \
    } while (0)

#define VS_TRY(expr)                                                                            \
    do {                                                                                        \
        int s_ = (expr);                                                                        \
        if (s_ != VS_OK) return s_;                                                             \
    } while (0)

// Owning device buffer; freed on destruction or reset.
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { reset(); }
    void reset()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    int alloc(size_t count)
    {
        reset();
        if (count == 0) count = 1;
        hipError_t e = hipMalloc((void **)&p, count * sizeof(T));
        if (e != hipSuccess) {
            p = nullptr;
            set_error("hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
            return e == hipErrorOutOfMemory ? VS_ERR_NOMEM : VS_ERR_HIP;
        }
        n = count;
        return VS_OK;
    }
    int upload(const T *src, size_t count, hipStream_t s)
    {
        if (count > n || !p) VS_TRY(alloc(count));
        if (count) VS_HIP(hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyDefault, s));
        return VS_OK;
    }
};

// Owning page-locked host buffer: the target of the small device-to-host copies inside a solve (a
// copy into pageable memory is staged by the runtime and costs several times as much).
template <typename T>
struct PinnedBuf {
    T *p = nullptr;
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf &) = delete;
    PinnedBuf &operator=(const PinnedBuf &) = delete;
    ~PinnedBuf() { if (p) (void)hipHostFree(p); }
    int alloc(size_t count)
    {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        hipError_t e = hipHostMalloc((void **)&p, count * sizeof(T), hipHostMallocDefault);
        if (e != hipSuccess) {
            p = nullptr;
            set_error("hipHostMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
            return VS_ERR_NOMEM;
        }
        return VS_OK;
    }
};

// Collective on a vs_comm (comm.hip): in-place sum or max of n doubles at a device pointer, enqueued
// on stream s (no host synchronisation).
int comm_allreduce(vs_comm *c, double *buf, int64_t n, bool take_max, hipStream_t s);
int comm_allreduce_min_u64(vs_comm *c, unsigned long long *buf, int64_t n, hipStream_t s);
int comm_rank(const vs_comm *c);
int comm_world(const vs_comm *c);

static inline int div_up(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

}  // namespace vs
