// em_common.cpp - thread-local error message of the C ABI (em_last_error).
#include "em_common.h"

#include <cstdarg>
#include <cstdio>

namespace {
thread_local char g_err[512] = "";
}

namespace em {
int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

int hip_fail(hipError_t e, const char* what)
{
    std::snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
    return EM_ERR_HIP;
}
}  // namespace em

extern "C" const char* em_last_error(void) { return g_err; }
