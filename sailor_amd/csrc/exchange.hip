// RCCL exchange step for split frames.  librccl is bound lazily so that the library loads (and the
// single-GPU path runs) on hosts without RCCL; the collective itself is recorded on the context stream.
#include "common.h"
#include <dlfcn.h>

typedef int (*nccl_allgather_fn)(const void*, void*, size_t, int /*ncclDataType_t*/, void* /*ncclComm_t*/, hipStream_t);

static nccl_allgather_fn resolve_allgather()
{
    static nccl_allgather_fn fn = nullptr;
    static bool tried = false;
    if (!tried) {
        tried = true;
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (h) fn = (nccl_allgather_fn)dlsym(h, "ncclAllGather");
    }
    return fn;
}

extern "C" int sailor_hip_allgather_u32(SailorHipContext* ctx, void* comm, const uint32_t* dSend, uint32_t* dRecv, size_t countPerRank)
{
    if (!ctx || !comm || !dSend || !dRecv) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    if (countPerRank == 0) return SAILOR_HIP_OK;
    nccl_allgather_fn fn = resolve_allgather();
    if (!fn) { ctx->lastError = "librccl.so not found"; return SAILOR_HIP_ERR_RCCL; }
    const int ncclUint32 = 3; // rccl.h: ncclInt8 0, ncclUint8 1, ncclInt32 2, ncclUint32 3
    const int rc = fn(dSend, dRecv, countPerRank, ncclUint32, comm, ctx->stream);
    if (rc != 0) { ctx->lastError = "ncclAllGather failed"; return SAILOR_HIP_ERR_RCCL; }
    return SAILOR_HIP_OK;
}
