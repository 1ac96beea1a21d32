// Error reporting and version entry points of the C ABI, and the process-wide memory pool.
#include <cstdarg>
#include <cstdio>
#include <mutex>
#include <vector>
#include "common.h"

namespace vs {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// Memory pool.  The reference builds a fresh Optimizer for every solve (src/core/tracking.cc:79-120),
// so handles are created and destroyed at frame rate; a hipMalloc / hipFree pair costs tens of
// microseconds and a handle owns some forty buffers.  Freed blocks of up to kPoolMaxBlock bytes are
// kept in per-device lists by power-of-two size class (at most kPoolMaxKept bytes per device and
// kind) and handed out again; larger blocks go straight to the runtime.
namespace {
constexpr size_t kPoolMinBlock = 256, kPoolMaxBlock = (size_t)64 << 20, kPoolMaxKept = (size_t)1 << 30;
constexpr int kPoolClasses = 19;      // 256 B << 18 = 64 MiB
constexpr int kPoolDevices = 64;
struct Pool {
    std::mutex m;
    std::vector<void *> free_list[kPoolDevices][kPoolClasses];
    size_t kept[kPoolDevices] = {};
};
// 0: device memory, 1: page-locked host memory, 2: page-locked coherent host memory.  Never
// destroyed: buffers may be let go of while the process winds down.
Pool *pools()
{
    static Pool *p = new Pool[3];
    return p;
}

int pool_class(size_t bytes)
{
    int c = 0;
    size_t sz = kPoolMinBlock;
    while (sz < bytes) { sz <<= 1; c++; }
    return c;
}
}  // namespace

// In a destructor that runs at process exit the runtime may already be gone: errors are ignored.
void *pool_alloc(size_t bytes, int kind, int *status, int *device)
{
    *status = VS_OK;
    if (bytes == 0) bytes = 1;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kPoolDevices) dev = -1;
    *device = dev;
    const bool host = kind != 0;
    Pool &pl = pools()[kind];
    size_t want = bytes;
    if (dev >= 0 && bytes <= kPoolMaxBlock) {
        const int c = pool_class(bytes);
        want = kPoolMinBlock << c;
        std::lock_guard<std::mutex> lock(pl.m);
        auto &fl = pl.free_list[dev][c];
        if (!fl.empty()) {
            void *p = fl.back();
            fl.pop_back();
            pl.kept[dev] -= want;
            return p;
        }
    }
    void *p = nullptr;
    const hipError_t e = kind == 2   ? hipHostMalloc(&p, want, hipHostMallocCoherent | hipHostMallocMapped)
                         : kind == 1 ? hipHostMalloc(&p, want, hipHostMallocDefault)
                                     : hipMalloc(&p, want);
    if (e != hipSuccess) {
        set_error("%s(%zu bytes) failed: %s", host ? "hipHostMalloc" : "hipMalloc", want, hipGetErrorString(e));
        *status = e == hipErrorOutOfMemory ? VS_ERR_NOMEM : VS_ERR_HIP;
        return nullptr;
    }
    return p;
}

void pool_free(void *p, size_t bytes, int kind, int dev)
{
    if (!p) return;
    if (bytes == 0) bytes = 1;
    const bool host = kind != 0;
    Pool &pl = pools()[kind];
    if (dev >= 0 && bytes <= kPoolMaxBlock) {
        const int c = pool_class(bytes);
        const size_t sz = kPoolMinBlock << c;
        std::lock_guard<std::mutex> lock(pl.m);
        if (pl.kept[dev] + sz <= kPoolMaxKept) {
            pl.free_list[dev][c].push_back(p);
            pl.kept[dev] += sz;
            return;
        }
    }
    if (host) (void)hipHostFree(p); else (void)hipFree(p);
}

}  // namespace vs

extern "C" {
const char *vs_last_error_string(void) { return vs::g_err; }
const char *vs_version(void) { return "vslam_hip 0.1 (gfx950)"; }
}
