// coala_internal.h -- shared by the translation units of libcoala_hip.so (not part of the ABI).
#ifndef COALA_INTERNAL_H
#define COALA_INTERNAL_H
#include <hip/hip_runtime.h>

// Records a thread-local message for coala_last_error() and returns `code`.
int coala_fail_(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

#define COALA_HIPCHK(expr)                                                                                   \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return coala_fail_(COALA_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,  \
                               __LINE__);                                                                    \
    } while (0)
#endif
