// The data-parallel exchange behind the C ABI (mnn_comm_init / mnn_allreduce_flat / mnn_comm_destroy; 8(e): ONE
// all-reduce(sum, f32) of the flat gradient buffer per step -- utils/training.py:151-177 computes the gradients of the whole variable list,
// which is that buffer here).  RCCL over xGMI; one communicator per process = per GPU.  The library never owns a communicator: the caller
// creates it, passes it to every call and destroys it, so "no global mutable state" holds.  A host binding that has no torch.distributed
// (the Python mirror uses it by default: multinn_amd/training.py) runs data parallel through these four calls alone; the 128-byte
// unique id travels from rank 0 to the other ranks through any channel the host has (a file, a socket, MPI, torch's store).
//
// RCCL is bound at RUN time (dlopen of librccl.so.1: the copy the process has already loaded -- PyTorch-ROCm brings its own -- or the
// system one), so libmultinn_hip.so has no link-time dependency on it and loads on a machine without RCCL; the comm calls then fail loudly.
#include "common.h"
#include <dlfcn.h>

namespace {
typedef struct { char internal[128]; } rccl_unique_id;          // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void* rccl_comm_t;
enum { RCCL_FLOAT32 = 7, RCCL_SUM = 0 };                          // ncclFloat32, ncclSum (rccl.h)
struct Rccl {
    int (*GetUniqueId)(rccl_unique_id*);
    int (*CommInitRank)(rccl_comm_t*, int, rccl_unique_id, int);
    int (*AllReduce)(const void*, void*, size_t, int, int, rccl_comm_t, hipStream_t);
    int (*CommDestroy)(rccl_comm_t);
    const char* (*GetErrorString)(int);
};
// resolved once per process; immutable afterwards (a function table, not state)
const Rccl* rccl() {
    static Rccl table;
    static const Rccl* ready = []() -> const Rccl* {
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);           // the copy already in the process (e.g. torch's)
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return nullptr;
        table.GetUniqueId = reinterpret_cast<int (*)(rccl_unique_id*)>(dlsym(h, "ncclGetUniqueId"));
        table.CommInitRank = reinterpret_cast<int (*)(rccl_comm_t*, int, rccl_unique_id, int)>(dlsym(h, "ncclCommInitRank"));
        table.AllReduce = reinterpret_cast<int (*)(const void*, void*, size_t, int, int, rccl_comm_t, hipStream_t)>(dlsym(h, "ncclAllReduce"));
        table.CommDestroy = reinterpret_cast<int (*)(rccl_comm_t)>(dlsym(h, "ncclCommDestroy"));
        table.GetErrorString = reinterpret_cast<const char* (*)(int)>(dlsym(h, "ncclGetErrorString"));
        if (!table.GetUniqueId || !table.CommInitRank || !table.AllReduce || !table.CommDestroy) return nullptr;
        return &table;
    }();
    return ready;
}
struct Comm { rccl_comm_t c; int rank, world; };
}  // namespace

#define MNN_RCCL(expr)                                                                                             \
    do {                                                                                                           \
        const int e_ = (expr);                                                                                     \
        if (e_ != 0) {                                                                                             \
            mnn_set_error("%s -> RCCL error %d (%s)", #expr, e_, R->GetErrorString ? R->GetErrorString(e_) : "?"); \
            return MNN_ERR_HIP;                                                                                    \
        }                                                                                                          \
    } while (0)

extern "C" int mnn_comm_unique_id(void* id_out) {
    MNN_REQUIRE(id_out != nullptr, "mnn_comm_unique_id: null pointer");
    const Rccl* R = rccl();
    const char* dle = R == nullptr ? dlerror() : nullptr;        // ONE call: dlerror() clears the message it returns
    MNN_REQUIRE(R != nullptr, "mnn_comm_unique_id: librccl.so.1 could not be loaded (%s)", dle ? dle : "symbols missing");
    rccl_unique_id id;
    MNN_RCCL(R->GetUniqueId(&id));
    memcpy(id_out, id.internal, sizeof(id.internal));
    return MNN_OK;
}

extern "C" int mnn_comm_init(mnn_comm_t* comm, int rank, int world, const void* id) {
    MNN_REQUIRE(comm && id && world >= 1 && rank >= 0 && rank < world, "mnn_comm_init: comm, id, 0 <= rank < world (rank %d of %d)", rank, world);
    const Rccl* R = rccl();
    MNN_REQUIRE(R != nullptr, "mnn_comm_init: librccl.so.1 could not be loaded");
    rccl_unique_id uid;
    memcpy(uid.internal, id, sizeof(uid.internal));
    Comm* c = new Comm{nullptr, rank, world};
    const int e = R->CommInitRank(&c->c, world, uid, rank);        // binds to the calling thread's current HIP device
    if (e != 0) {
        delete c;
        mnn_set_error("ncclCommInitRank(rank %d of %d) -> RCCL error %d (%s)", rank, world, e, R->GetErrorString ? R->GetErrorString(e) : "?");
        return MNN_ERR_HIP;
    }
    *comm = reinterpret_cast<mnn_comm_t>(c);
    return MNN_OK;
}

extern "C" int mnn_allreduce_flat(mnn_comm_t comm, mnn_stream_t s, float* buf, long n) {
    MNN_REQUIRE(comm && buf && n > 0, "mnn_allreduce_flat: comm, buffer and n > 0");
    const Rccl* R = rccl();
    MNN_REQUIRE(R != nullptr, "mnn_allreduce_flat: librccl.so.1 could not be loaded");
    Comm* c = reinterpret_cast<Comm*>(comm);
    MNN_RCCL(R->AllReduce(buf, buf, (size_t)n, RCCL_FLOAT32, RCCL_SUM, c->c, (hipStream_t)s));      // in place, asynchronous on the stream
    return MNN_OK;
}

extern "C" int mnn_comm_destroy(mnn_comm_t comm) {
    if (comm == nullptr) return MNN_OK;
    const Rccl* R = rccl();
    Comm* c = reinterpret_cast<Comm*>(comm);
    int rc = MNN_OK;
    if (R != nullptr && c->c != nullptr && R->CommDestroy(c->c) != 0) {
        mnn_set_error("ncclCommDestroy failed");
        rc = MNN_ERR_HIP;
    }
    delete c;
    return rc;
}
