// Communicator block of the C ABI (cpt_comm_init / cpt_allreduce_grads / cpt_allgather / cpt_comm_destroy), for hosts that are
// NOT PyTorch: the data-parallel step of cpt_amd itself keeps its collectives in torch.distributed (backend "nccl" = RCCL), driven from the
// bucket callbacks of cpt_train_fwd_ex / cpt_train_bwd_ex.  Replaces, for such a host, DistributedDataParallel's gradient all-reduce
// (/root/reference/Oscar/oscar/fewshot/refcoco_cpt.py:516-522) and the result gather of Oscar/oscar/utils/comm.py:102-142.
//
// RCCL is bound at RUN time (dlopen of its soname): libcpt_hip.so has no link-time dependency on it, a process that never calls these entry
// points never loads it, and inside a PyTorch process the soname resolves to the copy torch already loaded -- one RCCL per process.
// One communicator per process (one process drives one GPU, as torch.distributed launches the reference), created on the CURRENT device.
#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <mutex>

#include "common.h"
#include "cpt_hip.h"

namespace {

struct UniqueId { char internal[128]; };               // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void* Comm;                                    // ncclComm_t
enum { kSum = 0, kFloat32 = 7, kBfloat16 = 9 };        // ncclSum, ncclFloat32, ncclBfloat16 (rccl.h)

struct Api {
    void* handle = nullptr;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*ReduceScatter)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};

Api g_api;
Comm g_comm = nullptr;
int g_rank = -1, g_nranks = 0;
std::mutex g_mu;

}  // namespace

namespace cpt { int abi_fail(int code, const char* fmt, ...); }      // cpt_abi.hip: sets the calling thread's cpt_last_error() string
#define cfail cpt::abi_fail

namespace {

int load_api() {
    if (g_api.handle) return CPT_OK;
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) return cfail(CPT_ERR_ARCH, "cpt_comm: librccl.so.1 not found (%s)", dlerror());
#define SYM(field, name)                                                                       \
    do {                                                                                       \
        *(void**)(&g_api.field) = dlsym(h, name);                                               \
        if (!g_api.field) return cfail(CPT_ERR_ARCH, "cpt_comm: %s missing from librccl", name); \
    } while (0)
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(AllReduce, "ncclAllReduce");
    SYM(ReduceScatter, "ncclReduceScatter");
    SYM(AllGather, "ncclAllGather");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_api.handle = h;
    return CPT_OK;
}

int nccl_dtype(int dtype) { return dtype == CPT_F32 ? kFloat32 : (dtype == CPT_BF16 ? kBfloat16 : -1); }

int check_rc(int rc, const char* what) {
    if (rc == 0) return CPT_OK;
    return cfail(CPT_ERR_HIP, "%s: RCCL error %d (%s)", what, rc, g_api.GetErrorString ? g_api.GetErrorString(rc) : "?");
}

}  // namespace

extern "C" {

int cpt_comm_unique_id(void* id128) {
    if (!id128) return cfail(CPT_ERR_NULL, "cpt_comm_unique_id: null argument");
    std::lock_guard<std::mutex> lk(g_mu);
    if (int rc = load_api()) return rc;
    UniqueId id;
    if (int rc = check_rc(g_api.GetUniqueId(&id), "ncclGetUniqueId")) return rc;
    memcpy(id128, id.internal, sizeof(id.internal));
    return CPT_OK;
}

int cpt_comm_init(int rank, int nranks, const void* id128) {
    if (!id128) return cfail(CPT_ERR_NULL, "cpt_comm_init: null unique id");
    if (nranks <= 0 || rank < 0 || rank >= nranks) return cfail(CPT_ERR_SHAPE, "cpt_comm_init: rank %d of %d", rank, nranks);
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_comm) return cfail(CPT_ERR_SHAPE, "cpt_comm_init: this process already holds a communicator (rank %d of %d); cpt_comm_destroy first", g_rank, g_nranks);
    if (int rc = load_api()) return rc;
    UniqueId id;
    memcpy(id.internal, id128, sizeof(id.internal));
    Comm c = nullptr;
    if (int rc = check_rc(g_api.CommInitRank(&c, nranks, id, rank), "ncclCommInitRank")) return rc;
    g_comm = c; g_rank = rank; g_nranks = nranks;
    return CPT_OK;
}

int cpt_comm_rank(int* rank, int* nranks) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_comm) return cfail(CPT_ERR_NULL, "cpt_comm_rank: no communicator (cpt_comm_init)");
    if (rank) *rank = g_rank;
    if (nranks) *nranks = g_nranks;
    return CPT_OK;
}

int cpt_allreduce_grads(void* buf, size_t count, int dtype, void* stream) {
    const int dt = nccl_dtype(dtype);
    if (!buf) return cfail(CPT_ERR_NULL, "cpt_allreduce_grads: null buffer");
    if (dt < 0) return cfail(CPT_ERR_DTYPE, "cpt_allreduce_grads: dtype %d (CPT_F32 or CPT_BF16)", dtype);
    if (!g_comm) return cfail(CPT_ERR_NULL, "cpt_allreduce_grads: no communicator (cpt_comm_init)");
    return check_rc(g_api.AllReduce(buf, buf, count, dt, kSum, g_comm, (hipStream_t)stream), "ncclAllReduce");
}

int cpt_reduce_scatter(const void* send, void* recv, size_t recv_count, int dtype, void* stream) {
    const int dt = nccl_dtype(dtype);
    if (!send || !recv) return cfail(CPT_ERR_NULL, "cpt_reduce_scatter: null buffer");
    if (dt < 0) return cfail(CPT_ERR_DTYPE, "cpt_reduce_scatter: dtype %d (CPT_F32 or CPT_BF16)", dtype);
    if (!g_comm) return cfail(CPT_ERR_NULL, "cpt_reduce_scatter: no communicator (cpt_comm_init)");
    return check_rc(g_api.ReduceScatter(send, recv, recv_count, dt, kSum, g_comm, (hipStream_t)stream), "ncclReduceScatter");
}

int cpt_allgather(const void* send, void* recv, size_t send_count, int dtype, void* stream) {
    const int dt = nccl_dtype(dtype);
    if (!send || !recv) return cfail(CPT_ERR_NULL, "cpt_allgather: null buffer");
    if (dt < 0) return cfail(CPT_ERR_DTYPE, "cpt_allgather: dtype %d (CPT_F32 or CPT_BF16)", dtype);
    if (!g_comm) return cfail(CPT_ERR_NULL, "cpt_allgather: no communicator (cpt_comm_init)");
    return check_rc(g_api.AllGather(send, recv, send_count, dt, g_comm, (hipStream_t)stream), "ncclAllGather");
}

int cpt_comm_destroy(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_comm) return CPT_OK;
    const int rc = g_api.CommDestroy(g_comm);
    g_comm = nullptr; g_rank = -1; g_nranks = 0;
    return check_rc(rc, "ncclCommDestroy");
}

}  // extern "C"
