// Shared host-side helpers of libpleas_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>

#include "pleas_hip.h"

namespace pleas {

extern thread_local char g_last_error[256];

inline int hip_fail(hipError_t e, const char* what) {
    std::snprintf(g_last_error, sizeof(g_last_error), "%s: %s", what, hipGetErrorString(e));
    return PLEAS_EHIP;
}

inline int bad_arg(const char* what) {
    std::snprintf(g_last_error, sizeof(g_last_error), "invalid argument: %s", what);
    return PLEAS_EINVAL;
}

#define PLEAS_HIP_CHECK(expr)                                  \
    do {                                                       \
        hipError_t e_ = (expr);                                \
        if (e_ != hipSuccess) return ::pleas::hip_fail(e_, #expr); \
    } while (0)

#define PLEAS_LAUNCH_CHECK(name)                                   \
    do {                                                           \
        hipError_t e_ = hipGetLastError();                         \
        if (e_ != hipSuccess) return ::pleas::hip_fail(e_, name);  \
    } while (0)

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Pointers that reach a kernel through a device-side table are "generic" to hipcc, which then emits
// flat_load / flat_store: those count on lgkmcnt as well as vmcnt, so every LDS wait in the MFMA loop
// also waits for the global prefetch.  Casting to the global address space gives global_load / global_store.
typedef __attribute__((address_space(1))) float gfloat;
typedef __attribute__((address_space(1))) const float cgfloat;
typedef __attribute__((address_space(1))) const int gcint;
#define PLEAS_GLOBAL(p) ((::pleas::cgfloat*)(p))
#define PLEAS_GLOBAL_W(p) ((::pleas::gfloat*)(p))
#define PLEAS_GLOBAL_I(p) ((::pleas::gcint*)(p))

// ---- opt-in per-kernel timing with HIP events on the launch stream (pleas_prof_* in the C-ABI)
enum ProfKernel { kProfGramPartial = 0, kProfGramFinalize, kProfLsap, kProfMergeBlocks, kProfMaskedAdam, kProfSqerr,
                  kProfConvFwd, kProfConvWgrad, kProfNormalEq, kProfSolve, kProfBnAct, kProfCount };
extern bool g_prof_on;
extern unsigned g_prof_mask;   // bit k set: kernel id k is recorded while profiling is on
void prof_begin(int kernel, double flops, double bytes, hipStream_t stream);
void prof_end(hipStream_t stream);

struct ProfScope {
    hipStream_t s;
    bool on;
    ProfScope(int kernel, double flops, double bytes, hipStream_t stream) : s(stream), on(g_prof_on && ((g_prof_mask >> kernel) & 1u)) {
        if (on) prof_begin(kernel, flops, bytes, s);
    }
    ~ProfScope() {
        if (on) prof_end(s);
    }
};

}  // namespace pleas
