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

}  // namespace mccnn

extern "C" int mccnn_version(void) { return MCCNN_ABI_VERSION; }

extern "C" const char *mccnn_last_error_string(void) { return mccnn::g_err; }
