// The exchange step of data-parallel training behind the C boundary: an RCCL communicator and an in-place sum-all-reduce of
// a gradient range on the caller's stream (the reference: DistributedDataParallel's gradient all-reduce behind
// loss.backward(), train_gan.py:62-63,70-71 under the launcher of BASELINE config 3).  Host code only.
//
// RCCL is resolved at first use with dlopen / dlsym - libpfa_hip.so carries no link-time dependency on it, a one-GPU host
// never loads it, and inside a PyTorch process the library found under the soname librccl.so.1 is the one torch already
// mapped (one RCCL per process).  rccl.h supplies the types and enumerators only.
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>

#include <mutex>

#include "pfa_common.h"

struct pfa_comm {
    ncclComm_t comm;
    int nranks, rank;
};

namespace {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
} g_rccl;
std::mutex g_rccl_mutex;

int load_rccl() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return 0;
    // An RCCL the process has already mapped (torch brings its own copy) is reused: a second copy beside it would mean two
    // sets of communicator state and symbol interposition between them.  Only if none is mapped is one loaded, privately.
    void* h = nullptr;
    char tried[384] = "";
    static const char* const names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    for (const char* name : names) {
        h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
        if (h) break;
    }
    for (const char* name : names) {
        if (h) break;
        h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (!h) {
            const char* e = dlerror();
            const size_t used = strlen(tried);
            snprintf(tried + used, sizeof(tried) - used, "%s%s: %s", used ? "; " : "", name, e ? e : "?");
        }
    }
    if (!h) PFA_FAIL(-1, "pfa_comm: RCCL not found (%s)", tried);
    Rccl r;
    r.get_unique_id = (decltype(r.get_unique_id))dlsym(h, "ncclGetUniqueId");
    r.comm_init_rank = (decltype(r.comm_init_rank))dlsym(h, "ncclCommInitRank");
    r.comm_destroy = (decltype(r.comm_destroy))dlsym(h, "ncclCommDestroy");
    r.all_reduce = (decltype(r.all_reduce))dlsym(h, "ncclAllReduce");
    r.error_string = (decltype(r.error_string))dlsym(h, "ncclGetErrorString");
    if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_reduce || !r.error_string) {
        dlclose(h);
        PFA_FAIL(-1, "pfa_comm: the RCCL library lacks an entry point");
    }
    r.handle = h;
    g_rccl = r;
    return 0;
}

#define PFA_RCCL(call, what)                                                                     \
    do {                                                                                         \
        const ncclResult_t r__ = (call);                                                         \
        if (r__ != ncclSuccess) PFA_FAIL((int)r__, "%s: %s", what, g_rccl.error_string(r__));    \
    } while (0)

}  // namespace

extern "C" int pfa_comm_unique_id(void* out, int bytes) {
    PFA_REQUIRE(out && bytes == NCCL_UNIQUE_ID_BYTES, "pfa_comm_unique_id: the identifier is %d bytes", NCCL_UNIQUE_ID_BYTES);
    if (int rc = load_rccl()) return rc;
    ncclUniqueId id;
    PFA_RCCL(g_rccl.get_unique_id(&id), "ncclGetUniqueId");
    memcpy(out, id.internal, NCCL_UNIQUE_ID_BYTES);
    return 0;
}

extern "C" int pfa_comm_create(const void* unique_id, int bytes, int nranks, int rank, pfa_comm** out) {
    PFA_REQUIRE(unique_id && bytes == NCCL_UNIQUE_ID_BYTES && out && nranks >= 1 && rank >= 0 && rank < nranks,
                "pfa_comm_create: bad arguments");
    if (int rc = load_rccl()) return rc;
    ncclUniqueId id;
    memcpy(id.internal, unique_id, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t c = nullptr;
    PFA_RCCL(g_rccl.comm_init_rank(&c, nranks, id, rank), "ncclCommInitRank");      // (the calling thread's current device)
    *out = new pfa_comm{c, nranks, rank};
    return 0;
}

extern "C" int pfa_comm_destroy(pfa_comm* c) {
    if (!c) return 0;
    decltype(g_rccl.comm_destroy) destroy;
    {
        std::lock_guard<std::mutex> lock(g_rccl_mutex);
        PFA_REQUIRE(g_rccl.handle, "pfa_comm_destroy: no communicator was ever created");
        destroy = g_rccl.comm_destroy;
    }
    const ncclResult_t r = destroy(c->comm);
    delete c;
    if (r != ncclSuccess) PFA_FAIL((int)r, "ncclCommDestroy: %s", g_rccl.error_string(r));
    return 0;
}

extern "C" int pfa_comm_size(const pfa_comm* c) { return c ? c->nranks : 0; }
extern "C" int pfa_comm_rank(const pfa_comm* c) { return c ? c->rank : -1; }

extern "C" int pfa_allreduce(pfa_comm* c, void* buf, int64_t count, int dtype, pfa_stream s) {
    PFA_REQUIRE(c && count >= 0 && (buf || count == 0), "pfa_allreduce: bad arguments");
    PFA_REQUIRE(dtype == PFA_F32 || dtype == PFA_BF16, "pfa_allreduce: float32 or bfloat16 ranges");
    if (count == 0) return 0;
    PFA_RCCL(g_rccl.all_reduce(buf, buf, (size_t)count, dtype == PFA_F32 ? ncclFloat32 : ncclBfloat16, ncclSum, c->comm,
                               (hipStream_t)s),
             "ncclAllReduce");
    return 0;
}
