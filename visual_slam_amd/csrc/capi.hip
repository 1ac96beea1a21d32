// Error reporting and version entry points of the C ABI.
#include <cstdarg>
#include <cstdio>
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
}  // namespace vs

extern "C" {
const char *vs_last_error_string(void) { return vs::g_err; }
const char *vs_version(void) { return "vslam_hip 0.1 (gfx950)"; }
}
