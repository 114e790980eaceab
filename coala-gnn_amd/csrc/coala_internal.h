// coala_internal.h -- shared by the translation units of libcoala_hip.so (not part of the ABI).
#ifndef COALA_INTERNAL_H
#define COALA_INTERNAL_H
#include <hip/hip_runtime.h>
#include <stdint.h>

// Records a thread-local message for coala_last_error() and returns `code`.
int coala_fail_(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

#define COALA_HIPCHK(expr)                                                                                   \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return coala_fail_(COALA_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,  \
                               __LINE__);                                                                    \
    } while (0)

// The split-phase serve calls with an event ON a launch instead of behind it (coala_cache.hip; used by the distributed fetch, coala_comm.cpp):
// begin_ev rides on the probe's launch, end_ev on the last fill launch of the call.  *rode = the event that really rides there: the caller's, or
// the handle's own profiling event of that launch (a profiling handle keeps the dispatches' event slots: fine to wait on, not to time with), or
// NULL when the call launched nothing (the caller then records its event the plain way).
struct coala_cache;
struct coala_row_redirect;
extern "C" {
int coala_serve_probe_redirect_ev_(coala_cache* h, float* out, const int64_t* ids, int64_t n, const coala_row_redirect* redirect, void* stream,
                                   hipEvent_t begin_ev, hipEvent_t* rode);
int coala_serve_fill_ranges_ev_(coala_cache* h, float* out, const int64_t* ids, int64_t n, const int64_t* begins, const int64_t* ends, int n_ranges,
                                void* stream, hipEvent_t end_ev, hipEvent_t* rode);
}
#endif
