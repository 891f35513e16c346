// lvba_common.h -- error reporting, HIP/RCCL call checks shared by the host-side translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h> // types/enums only; librccl.so is dlopen()ed on first multi-GPU use
#include <stdint.h>
#include "../../include/lvba_hip.h"

// Sets the thread-local message returned by lvba_last_error() and returns `code`.
int32_t lvba_fail(int32_t code, const char *fmt, ...);

#define HIPCHK(expr)                                                                                        \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess)                                                                               \
            return lvba_fail(e_ == hipErrorOutOfMemory ? LVBA_ERR_NOMEM : LVBA_ERR_DEVICE, "%s: %s (%s:%d)", \
                             #expr, hipGetErrorString(e_), __FILE__, __LINE__);                             \
    } while (0)
#define TRY(expr) do { int32_t rc_ = (expr); if (rc_ != LVBA_OK) return rc_; } while (0)

struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
extern RcclApi g_rccl;
int32_t rccl_load();
#define NCCLCHK(expr)                                                                                       \
    do {                                                                                                    \
        ncclResult_t r_ = (expr);                                                                           \
        if (r_ != ncclSuccess)                                                                              \
            return lvba_fail(LVBA_ERR_DIST, "%s: %s", #expr, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "rccl error"); \
    } while (0)
