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

// Blocks from the process-wide pool (capi.hip): device memory, or page-locked host memory.  A block
// belongs to the device that was current at pool_alloc (*device; -1: not pooled); `bytes` and
// `device` at pool_free are those of pool_alloc.
// kind: 0 device memory, 1 page-locked host memory, 2 page-locked host memory that is coherent
// between host and GPU while a kernel runs (a kernel's stores become visible to a polling host).
void *pool_alloc(size_t bytes, int kind, int *status, int *device);
void pool_free(void *p, size_t bytes, int kind, int device);

// Owning device buffer; returned to the pool on destruction or reset.  The owner makes sure no
// work that uses the buffer is still running when it lets go of it: handles synchronise their
// stream before they are destroyed, and growing a live buffer waits for the device first.
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    int dev = -1;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    bool is_view = false;    // points into another buffer (see view()): nothing to give back
    void release()
    {
        if (p && !is_view) pool_free(p, n * sizeof(T), 0, dev);
        p = nullptr;
        n = 0;
        is_view = false;
    }
    // Non-owning window of `count` elements at `ptr` (part of an arena that outlives this buffer).
    void view(T *ptr, size_t count)
    {
        release();
        p = ptr; n = count; is_view = true;
    }
    void reset()
    {
        if (p && !is_view) (void)hipDeviceSynchronize();     // hipFree would have waited as well
        release();
    }
    int alloc(size_t count)
    {
        reset();
        if (count == 0) count = 1;
        int rc = VS_OK;
        p = (T *)pool_alloc(count * sizeof(T), 0, &rc, &dev);
        if (rc != VS_OK) { p = nullptr; return rc; }
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
template <typename T, bool COHERENT = false>
struct PinnedBuf {
    static constexpr int kKind = COHERENT ? 2 : 1;
    T *p = nullptr;
    size_t n = 0;
    int dev = -1;
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf &) = delete;
    PinnedBuf &operator=(const PinnedBuf &) = delete;
    ~PinnedBuf() { if (p) pool_free(p, n * sizeof(T), kKind, dev); }
    int alloc(size_t count)
    {
        if (p) pool_free(p, n * sizeof(T), kKind, dev);
        p = nullptr; n = 0;
        int rc = VS_OK;
        p = (T *)pool_alloc(count * sizeof(T), kKind, &rc, &dev);
        if (rc != VS_OK) { p = nullptr; return rc; }
        n = count;
        return VS_OK;
    }
};

// Collective on a vs_comm (comm.hip): in-place sum or max of n doubles at a device pointer, enqueued
// on stream s (no host synchronisation).
int comm_allreduce(vs_comm *c, double *buf, int64_t n, bool take_max, hipStream_t s);
int comm_allreduce_min_u64(vs_comm *c, unsigned long long *buf, int64_t n, hipStream_t s);
int comm_allgather(vs_comm *c, const double *send, double *recv, int64_t n, hipStream_t s);
int comm_rank(const vs_comm *c);
int comm_world(const vs_comm *c);

static inline int div_up(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

}  // namespace vs
