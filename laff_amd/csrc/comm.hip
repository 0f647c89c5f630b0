// comm.hip -- the collectives of the sharded path behind the C ABI (include/laff_hip.h, "e: multi-GPU"), for a host that is not
// Python: laff_amd/dist.py issues the same three collectives through torch.distributed.  RCCL is NOT linked: it is looked up at the
// first call -- a copy the process has already loaded (torch ships one) is taken before a new one is opened, so that one process never
// runs two RCCL instances by accident.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#if __has_include(<rccl/rccl.h>) && !defined(LAFF_NO_RCCL_HEADER)
#include <rccl/rccl.h>
#else
// A ROCm install without the RCCL headers: the library is looked up at run time anyway, so the handful of types and enumerators of
// NCCL's stable C ABI that this file uses are declared here (values as in nccl.h / rccl.h).
extern "C" {
typedef struct ncclComm* ncclComm_t;
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7,
               ncclFloat64 = 8, ncclBfloat16 = 9 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3, ncclAvg = 4 } ncclRedOp_t;
}
#endif

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/laff_hip.h"

extern "C" int laff_ctx_stream_device(laff_ctx* ctx, void** stream, int* device);      // api.hip
extern "C" void laff_set_error(const char* msg);                                       // api.hip

struct laff_comm {
    ncclComm_t comm;
    hipStream_t stream;
    int device, rank, world;
};

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int failf(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    laff_set_error(buf);
    return code;
}

int load_rccl_once() {
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;              // a copy that is already in the process
    for (int i = 0; !h && i < 3; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!h) return failf(LAFF_E_UNSUPPORTED, "laff_comm: cannot load RCCL (librccl.so / librccl.so.1): %s", dlerror());
    Rccl r;
    r.handle = h;
#define LAFF_SYM(field, name)                                                                        \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name));                                   \
    if (!r.field) return failf(LAFF_E_UNSUPPORTED, "laff_comm: RCCL has no symbol %s", name)
    LAFF_SYM(GetUniqueId, "ncclGetUniqueId");
    LAFF_SYM(CommInitRank, "ncclCommInitRank");
    LAFF_SYM(CommDestroy, "ncclCommDestroy");
    LAFF_SYM(AllGather, "ncclAllGather");
    LAFF_SYM(AllReduce, "ncclAllReduce");
    LAFF_SYM(GetErrorString, "ncclGetErrorString");
#undef LAFF_SYM
    g_rccl = r;
    return LAFF_OK;
}

// first callers may race: one of them loads, the others wait; a failed load is retried by the next call (its message is thread-local)
int load_rccl() {
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (g_rccl.handle) return LAFF_OK;
    return load_rccl_once();
}

#define RCCL_TRY(expr)                                                                                         \
    do {                                                                                                       \
        ncclResult_t r_ = (expr);                                                                              \
        if (r_ != ncclSuccess) return failf(LAFF_E_HIP, "%s: %s", #expr, g_rccl.GetErrorString(r_));           \
    } while (0)

struct DevGuard {
    int prev = -1;
    explicit DevGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
    }
    ~DevGuard() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};

}  // namespace

extern "C" {

static_assert(LAFF_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id travels as RCCL's own unique id");

int laff_comm_unique_id(unsigned char* id) {
    if (!id) return failf(LAFF_E_ARG, "laff_comm_unique_id: null id");
    if (int rc = load_rccl()) return rc;
    ncclUniqueId u;
    RCCL_TRY(g_rccl.GetUniqueId(&u));
    memcpy(id, u.internal, LAFF_COMM_ID_BYTES);
    return LAFF_OK;
}

int laff_comm_init(laff_ctx* ctx, int rank, int world, const unsigned char* id, laff_comm** out) {
    if (!ctx || !id || !out) return failf(LAFF_E_ARG, "laff_comm_init: null argument");
    if (world < 1 || rank < 0 || rank >= world) return failf(LAFF_E_ARG, "laff_comm_init: rank %d of %d", rank, world);
    if (int rc = load_rccl()) return rc;
    void* st = nullptr;
    int dev = 0;
    if (int rc = laff_ctx_stream_device(ctx, &st, &dev)) return rc;
    DevGuard g(dev);
    ncclUniqueId u;
    memcpy(u.internal, id, LAFF_COMM_ID_BYTES);
    ncclComm_t c;
    RCCL_TRY(g_rccl.CommInitRank(&c, world, u, rank));
    *out = new laff_comm{c, static_cast<hipStream_t>(st), dev, rank, world};
    return LAFF_OK;
}

int laff_comm_set_stream(laff_comm* comm, void* hip_stream) {
    if (!comm) return failf(LAFF_E_ARG, "laff_comm_set_stream: null comm");
    comm->stream = static_cast<hipStream_t>(hip_stream);
    return LAFF_OK;
}

int laff_comm_destroy(laff_comm* comm) {
    if (!comm) return LAFF_OK;
    DevGuard g(comm->device);
    ncclResult_t r = g_rccl.CommDestroy ? g_rccl.CommDestroy(comm->comm) : ncclSuccess;
    delete comm;
    if (r != ncclSuccess) return failf(LAFF_E_HIP, "ncclCommDestroy: %s", g_rccl.GetErrorString(r));
    return LAFF_OK;
}

int laff_allgather_rows(laff_comm* comm, const void* send, void* recv, size_t bytes_per_rank) {
    if (!comm) return failf(LAFF_E_ARG, "laff_allgather_rows: null comm");
    if (bytes_per_rank == 0) return LAFF_OK;
    if (!send || !recv) return failf(LAFF_E_ARG, "laff_allgather_rows: null buffer");
    DevGuard g(comm->device);
    RCCL_TRY(g_rccl.AllGather(send, recv, bytes_per_rank, ncclChar, comm->comm, comm->stream));
    return LAFF_OK;
}

int laff_allreduce_i32_sum(laff_comm* comm, int* buf, size_t n) {
    if (!comm) return failf(LAFF_E_ARG, "laff_allreduce_i32_sum: null comm");
    if (n == 0) return LAFF_OK;
    if (!buf) return failf(LAFF_E_ARG, "laff_allreduce_i32_sum: null buffer");
    DevGuard g(comm->device);
    RCCL_TRY(g_rccl.AllReduce(buf, buf, n, ncclInt32, ncclSum, comm->comm, comm->stream));
    return LAFF_OK;
}

int laff_allreduce_f64_max(laff_comm* comm, double* buf, size_t n) {
    if (!comm) return failf(LAFF_E_ARG, "laff_allreduce_f64_max: null comm");
    if (n == 0) return LAFF_OK;
    if (!buf) return failf(LAFF_E_ARG, "laff_allreduce_f64_max: null buffer");
    DevGuard g(comm->device);
    RCCL_TRY(g_rccl.AllReduce(buf, buf, n, ncclFloat64, ncclMax, comm->comm, comm->stream));
    return LAFF_OK;
}

}  // extern "C"
