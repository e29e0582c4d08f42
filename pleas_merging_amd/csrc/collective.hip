// pleas_allreduce_sum: the path's exchange step -- the sum over ranks of a flat fp32 arena (the 41 MB cost arena of
// activation matching before the LAPs; the A / B arenas of the closed form before the solve; the gradient arena of an
// Adam update) -- as ONE RCCL all-reduce over xGMI, for consumers of the C-ABI that do not run under torch.distributed.
//
// There is no reference counterpart: the reference is single-GPU (its only multi-GPU line is an unused
// nn.DataParallel, experiments/datasets/common.py:68); SURVEY.md section 8(b) lists the entry point, section 8(e) the
// partitioning.  The Python host side keeps using torch.distributed (backend "nccl" IS RCCL), whose communicator cannot
// be handed out; this entry point takes the CALLER's ncclComm_t.
//
// RCCL is not linked: a communicator is only valid inside the library instance that created it, so the symbol is taken
// from the instance already loaded in the process (dlsym(RTLD_DEFAULT)), else from $PLEAS_RCCL_LIB, else from
// librccl.so.1 on the loader path.
#include <dlfcn.h>

#include <mutex>

#include "common.hpp"

namespace pleas {

typedef int (*nccl_all_reduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*nccl_error_string_fn)(int);
constexpr int kNcclFloat32 = 7, kNcclSum = 0;   // rccl.h: ncclDataType_t / ncclRedOp_t

static nccl_all_reduce_fn g_all_reduce = nullptr;
static nccl_error_string_fn g_error_string = nullptr;
static std::once_flag g_rccl_once;

static void load_rccl() {
    void* sym = dlsym(RTLD_DEFAULT, "ncclAllReduce");
    void* handle = nullptr;
    if (!sym) {
        const char* path = std::getenv("PLEAS_RCCL_LIB");
        handle = dlopen(path && path[0] ? path : "librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!handle && !(path && path[0])) handle = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (handle) sym = dlsym(handle, "ncclAllReduce");
    }
    g_all_reduce = reinterpret_cast<nccl_all_reduce_fn>(sym);
    void* es = handle ? dlsym(handle, "ncclGetErrorString") : dlsym(RTLD_DEFAULT, "ncclGetErrorString");
    g_error_string = reinterpret_cast<nccl_error_string_fn>(es);
}

}  // namespace pleas

using namespace pleas;

extern "C" int pleas_allreduce_sum(float* buf, int64_t n, void* comm, void* stream) {
    if (!buf || !comm) return bad_arg("allreduce_sum: null buffer / communicator");
    if (n < 0) return bad_arg("allreduce_sum: negative count");
    if (n == 0) return PLEAS_OK;
    std::call_once(g_rccl_once, load_rccl);
    if (!g_all_reduce) {
        std::snprintf(g_last_error, sizeof(g_last_error), "allreduce_sum: RCCL (ncclAllReduce) not found in the process, "
                      "$PLEAS_RCCL_LIB or librccl.so.1");
        return PLEAS_EHIP;
    }
    const int rc = g_all_reduce(buf, buf, (size_t)n, kNcclFloat32, kNcclSum, comm, (hipStream_t)stream);   // in place
    if (rc != 0) {
        std::snprintf(g_last_error, sizeof(g_last_error), "ncclAllReduce: %s", g_error_string ? g_error_string(rc) : "error");
        return PLEAS_EHIP;
    }
    return PLEAS_OK;
}
