// Ensemble diagnostics gather over RCCL, without torch (replicas are independent; the
// only exchange is an all-gather of a few float64 scalars per output chunk).  The reference has no
// distributed code, so this row has no reference counterpart; the torch.distributed route of
// quflow_amd/ensemble.py stays the default and this one is the torch-free alternative (one process per GPU,
// ncclCommInitRank over a 128-byte id the ranks exchange themselves).
//
// librccl.so is opened lazily with dlopen so that libquflow_hip.so has no link-time dependency on it:
// a single-GPU user never loads it.
#include "qf_internal.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>

namespace {

struct rccl_api {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

rccl_api g_rccl;

int load_rccl()
{
    if (g_rccl.handle) return QF_OK;
    const char *names[] = {getenv("QUFLOW_HIP_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names) {
        if (!n || !*n) continue;
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) {
        qf_set_error("qf_comm: librccl.so not found (%s); set QUFLOW_HIP_RCCL_LIB", dlerror());
        return QF_ERR_STATE;
    }
    rccl_api a;
    a.handle = h;
#define QF_SYM(field, name)                                                   \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, name));            \
    if (!a.field) {                                                           \
        qf_set_error("qf_comm: %s missing from librccl", name);               \
        dlclose(h);                                                           \
        return QF_ERR_STATE;                                                  \
    }
    QF_SYM(GetUniqueId, "ncclGetUniqueId")
    QF_SYM(CommInitRank, "ncclCommInitRank")
    QF_SYM(CommDestroy, "ncclCommDestroy")
    QF_SYM(AllGather, "ncclAllGather")
    QF_SYM(AllReduce, "ncclAllReduce")
    QF_SYM(GetErrorString, "ncclGetErrorString")
#undef QF_SYM
    g_rccl = a;
    return QF_OK;
}

#define QF_RCCL(call)                                                                         \
    do {                                                                                      \
        ncclResult_t _r = (call);                                                             \
        if (_r != ncclSuccess) {                                                              \
            qf_set_error("%s failed: %s (%s:%d)", #call, g_rccl.GetErrorString(_r), __FILE__, __LINE__); \
            return QF_ERR_HIP;                                                                \
        }                                                                                     \
    } while (0)

}  // namespace

struct qf_comm {
    int device = 0, nranks = 1, rank = 0;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    double *send_dev = nullptr, *recv_dev = nullptr;
    size_t capacity = 0;  // doubles per rank the staging buffers hold
};

static int comm_reserve(qf_comm *c, size_t count)
{
    if (count <= c->capacity) return QF_OK;
    size_t cap = c->capacity ? c->capacity : 64;
    while (cap < count) cap *= 2;
    if (c->send_dev) QF_HIP(hipFree(c->send_dev));
    if (c->recv_dev) QF_HIP(hipFree(c->recv_dev));
    c->send_dev = c->recv_dev = nullptr;
    c->capacity = 0;
    QF_HIP(hipMalloc(&c->send_dev, cap * sizeof(double)));
    QF_HIP(hipMalloc(&c->recv_dev, cap * sizeof(double) * c->nranks));
    c->capacity = cap;
    return QF_OK;
}

extern "C" {

int qf_comm_unique_id(void *id128)
{
    if (!id128) {
        qf_set_error("qf_comm_unique_id: null output");
        return QF_ERR_INVALID;
    }
    QF_TRY(load_rccl());
    ncclUniqueId id;
    QF_RCCL(g_rccl.GetUniqueId(&id));
    static_assert(sizeof(id) == QF_COMM_ID_BYTES, "unique id size");
    memcpy(id128, &id, sizeof(id));
    return QF_OK;
}

int qf_comm_create(qf_comm **out, int device, int nranks, int rank, const void *id128)
{
    if (!out || !id128) {
        qf_set_error("qf_comm_create: null argument");
        return QF_ERR_INVALID;
    }
    *out = nullptr;
    if (nranks < 1 || rank < 0 || rank >= nranks) {
        qf_set_error("qf_comm_create: rank %d of %d", rank, nranks);
        return QF_ERR_INVALID;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        qf_set_error("qf_comm_create: no HIP device visible");
        return QF_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= ndev) {
        qf_set_error("qf_comm_create: device %d out of range (0..%d)", device, ndev - 1);
        return QF_ERR_NO_DEVICE;
    }
    QF_TRY(load_rccl());
    QF_HIP(hipSetDevice(device));
    qf_comm *c = new qf_comm;
    c->device = device;
    c->nranks = nranks;
    c->rank = rank;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) {
        qf_set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, nranks, g_rccl.GetErrorString(r));
        delete c;
        return QF_ERR_HIP;
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || comm_reserve(c, 64) != QF_OK) {
        qf_set_error("qf_comm_create: stream / staging allocation failed");
        g_rccl.CommDestroy(c->comm);
        delete c;
        return QF_ERR_HIP;
    }
    *out = c;
    return QF_OK;
}

int qf_comm_allgather_f64(qf_comm *c, const double *send_host, int count, double *recv_host)
{
    if (!c || count < 0 || (count > 0 && (!send_host || !recv_host))) {
        qf_set_error("qf_comm_allgather_f64: bad argument");
        return QF_ERR_INVALID;
    }
    if (count == 0) return QF_OK;
    QF_HIP(hipSetDevice(c->device));
    QF_TRY(comm_reserve(c, (size_t)count));
    QF_HIP(hipMemcpyAsync(c->send_dev, send_host, sizeof(double) * count, hipMemcpyHostToDevice, c->stream));
    QF_RCCL(g_rccl.AllGather(c->send_dev, c->recv_dev, (size_t)count, ncclDouble, c->comm, c->stream));
    QF_HIP(hipMemcpyAsync(recv_host, c->recv_dev, sizeof(double) * count * c->nranks, hipMemcpyDeviceToHost, c->stream));
    QF_HIP(hipStreamSynchronize(c->stream));
    return QF_OK;
}

int qf_comm_barrier(qf_comm *c)
{
    if (!c) {
        qf_set_error("qf_comm_barrier: null comm");
        return QF_ERR_INVALID;
    }
    QF_HIP(hipSetDevice(c->device));
    QF_HIP(hipMemsetAsync(c->send_dev, 0, sizeof(double), c->stream));
    QF_RCCL(g_rccl.AllReduce(c->send_dev, c->recv_dev, 1, ncclDouble, ncclSum, c->comm, c->stream));
    QF_HIP(hipStreamSynchronize(c->stream));
    return QF_OK;
}

int qf_comm_destroy(qf_comm *c)
{
    if (!c) return QF_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)g_rccl.CommDestroy(c->comm);
    if (c->send_dev) (void)hipFree(c->send_dev);
    if (c->recv_dev) (void)hipFree(c->recv_dev);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return QF_OK;
}

}  // extern "C"
