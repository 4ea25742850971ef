// Multi-GPU merge step of the C-ABI (include/bsx.h, bsx_comm_*): one RCCL communicator per handle
// (= per rank = per GPU) and a byte all-gather on the engine's stream.  Replaces the result collection
// of the reference's mpi4py task farm (boolsi/mpi.py:290-330: the master receives every worker's batch
// results and writes them to the database): here the problems are range-partitioned, so the only
// exchange is ONE all-gather of the per-rank attractor tables at the end of a run.
//
// librccl.so (570 MB) is loaded on the first bsx_comm_* call, not with this library: single-GPU runs
// never touch it.  The ncclUniqueId travels between the ranks through the caller (boolsi_amd/dist.py
// passes it over its TCP bootstrap); nothing here opens a socket.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

#include "bsx_engine.h"

namespace {

struct Rccl {
    void* lib = nullptr;
    std::string error;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

Rccl g_rccl;
std::once_flag g_rccl_once;

const Rccl& rccl() {
    std::call_once(g_rccl_once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            g_rccl.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (g_rccl.lib) break;
        }
        if (!g_rccl.lib) {
            const char* why = dlerror();
            g_rccl.error = std::string("cannot load librccl.so: ") + (why ? why : "not found");
            return;
        }
        auto sym = [](const char* n) { return dlsym(g_rccl.lib, n); };
        g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(sym("ncclGetUniqueId"));
        g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(sym("ncclCommInitRank"));
        g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(sym("ncclCommDestroy"));
        g_rccl.AllGather = reinterpret_cast<decltype(g_rccl.AllGather)>(sym("ncclAllGather"));
        g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
        if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllGather || !g_rccl.GetErrorString)
            g_rccl.error = "librccl.so lacks a required ncclXxx symbol";
    });
    return g_rccl;
}

int nccl_fail(bsx_handle h, const char* what, ncclResult_t r) {
    return fail(h, BSX_ERR_COMM, std::string(what) + ": " + rccl().GetErrorString(r));
}

}  // namespace

#define NCCLCHK(h, call)                                                  \
    do {                                                                  \
        ncclResult_t r_ = (call);                                         \
        if (r_ != ncclSuccess) return nccl_fail((h), #call, r_);          \
    } while (0)

extern "C" int bsx_comm_unique_id(bsx_handle h, void* out, uint32_t cap) {
    if (!h) return BSX_ERR_INVALID;
    if (!out || cap < BSX_COMM_ID_BYTES) return fail(h, BSX_ERR_INVALID, "bsx_comm_unique_id: buffer smaller than BSX_COMM_ID_BYTES");
    static_assert(sizeof(ncclUniqueId) == BSX_COMM_ID_BYTES, "ncclUniqueId size");
    const Rccl& R = rccl();
    if (!R.error.empty()) return fail(h, BSX_ERR_COMM, R.error);
    HIPCHK(h, hipSetDevice(h->device));
    ncclUniqueId id;
    NCCLCHK(h, R.GetUniqueId(&id));
    std::memcpy(out, &id, sizeof(id));
    return BSX_OK;
}

extern "C" int bsx_comm_init(bsx_handle h, const void* unique_id, uint32_t id_bytes, int rank, int world) {
    if (!h) return BSX_ERR_INVALID;
    if (!unique_id || id_bytes != BSX_COMM_ID_BYTES) return fail(h, BSX_ERR_INVALID, "bsx_comm_init: unique id must be BSX_COMM_ID_BYTES bytes");
    if (world < 1 || rank < 0 || rank >= world) return fail(h, BSX_ERR_INVALID, "bsx_comm_init: rank outside [0, world)");
    if (h->comm) return fail(h, BSX_ERR_STATE, "bsx_comm_init: the handle already has a communicator");
    const Rccl& R = rccl();
    if (!R.error.empty()) return fail(h, BSX_ERR_COMM, R.error);
    HIPCHK(h, hipSetDevice(h->device));
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    ncclComm_t comm = nullptr;
    NCCLCHK(h, R.CommInitRank(&comm, world, id, rank));
    h->comm = comm;
    h->comm_rank = rank;
    h->comm_world = world;
    return BSX_OK;
}

extern "C" int bsx_comm_allgather(bsx_handle h, const void* send, uint64_t bytes_per_rank, void* recv) {
    if (!h) return BSX_ERR_INVALID;
    if (!h->comm) return fail(h, BSX_ERR_STATE, "bsx_comm_allgather before bsx_comm_init");
    if (bytes_per_rank == 0) return BSX_OK;
    if (!send || !recv) return fail(h, BSX_ERR_INVALID, "bsx_comm_allgather: null buffer");
    const Rccl& R = rccl();
    HIPCHK(h, hipSetDevice(h->device));
    const size_t total = (size_t)bytes_per_rank * (size_t)h->comm_world;
    HIPCHK(h, h->d_comm_send.reserve(bytes_per_rank));
    HIPCHK(h, h->d_comm_recv.reserve(total));
    HIPCHK(h, hipMemcpyAsync(h->d_comm_send.p, send, bytes_per_rank, hipMemcpyHostToDevice, h->stream));
    NCCLCHK(h, R.AllGather(h->d_comm_send.p, h->d_comm_recv.p, bytes_per_rank, ncclUint8, static_cast<ncclComm_t>(h->comm), h->stream));
    HIPCHK(h, hipMemcpyAsync(recv, h->d_comm_recv.p, total, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return BSX_OK;
}

extern "C" int bsx_comm_destroy(bsx_handle h) {
    if (!h) return BSX_ERR_INVALID;
    if (!h->comm) return BSX_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    const ncclResult_t r = rccl().CommDestroy(static_cast<ncclComm_t>(h->comm));
    h->comm = nullptr;
    h->comm_world = 1;
    h->comm_rank = 0;
    if (r != ncclSuccess) return nccl_fail(h, "ncclCommDestroy", r);
    return BSX_OK;
}
