// Error reporting and version entry points of the C ABI (include/mccnn.h).
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace mccnn {

static thread_local char g_err[512] = "no error";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return 0;
    set_error("%s: %s", what, hipGetErrorString(e));
    return (int)e;
}

int device_cus8()
{
    static int cache[64] = {0};   // benign race: every thread computes the same value
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cache[dev] == 0) {
        int c = 0;
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256;
        c &= ~7;
        cache[dev] = c < 8 ? 8 : c;
    }
    return cache[dev];
}

}  // namespace mccnn

extern "C" int mccnn_version(void) { return MCCNN_ABI_VERSION; }

extern "C" const char *mccnn_last_error_string(void) { return mccnn::g_err; }
