// em_common.h - error plumbing shared by the translation units of libedgemarl.so.
#pragma once
#include <hip/hip_runtime.h>

#include "edgemarl.h"

namespace em {
// Stores a printf-formatted message in the calling thread's error slot and returns `code`.
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
int hip_fail(hipError_t e, const char* what);
inline int hip_check(hipError_t e, const char* what) { return e == hipSuccess ? EM_OK : hip_fail(e, what); }
}  // namespace em

#define EM_HIP(call)                                              \
    do {                                                          \
        hipError_t em_hip_e_ = (call);                            \
        if (em_hip_e_ != hipSuccess) return em::hip_fail(em_hip_e_, #call); \
    } while (0)
