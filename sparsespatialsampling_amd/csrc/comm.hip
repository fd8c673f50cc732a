// Multi-GPU exchange steps of the S^3 path inside the library: RCCL over xGMI, one communicator per process (one process
// per GPU).  gfx950 only.
//
// The path shards without a data-path collective for the interpolation (leaf cells are split into contiguous ranges,
// every rank interpolates its own range).  The level-synchronous refine has one real exchange per batch: every rank
// evaluates the KNN metric / gain of its 1/W slice of the new cells (the reference does this with a process pool and
// pickled tuples, s_cube.py:207-241), one grouped all-gather returns the slices to everybody.  That is the only exchange of a
// refinement step (r5): the captured metric (s_cube.py:317-336) is reduced from the then replicated arrays on every rank, per
// fixed 1024-cell block and in block order -- the result does not depend on the number of ranks.
//
// RCCL is looked up at run time (dlopen of the soname torch has already loaded, so that both use the same copy);
// libs3hip.so itself has no link-time dependency on it.
#include "common.h"

#include <dlfcn.h>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
// An image without the RCCL headers still builds the library (single-GPU use needs none of this); the few types and
// constants of the calls below, as the RCCL / NCCL ABI defines them.  The functions are looked up at run time either way.
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6,
               ncclFloat32 = 7, ncclFloat64 = 8, ncclDouble = 8 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;
}
#endif

#include <cstring>

struct s3_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl *rccl() {
    static Rccl r;
    static bool tried = false;
    if (tried) return r.lib ? &r : nullptr;
    tried = true;
    for (const char *name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
        r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (r.lib) break;
    }
    if (!r.lib) return nullptr;
    bool ok = true;
    auto sym = [&](const char *n) {
        void *p = dlsym(r.lib, n);
        ok = ok && p != nullptr;
        return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
    r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    if (!ok) {
        r.lib = nullptr;
        return nullptr;
    }
    return &r;
}

#define S3_RCCL(expr)                                                                                  \
    do {                                                                                               \
        ncclResult_t _r = (expr);                                                                      \
        if (_r != ncclSuccess) {                                                                       \
            s3::set_error("%s failed: %s (%s:%d)", #expr, R->GetErrorString(_r), __FILE__, __LINE__);  \
            return S3_EHIP;                                                                            \
        }                                                                                              \
    } while (0)

}  // namespace

extern "C" {

// 1 when librccl can be loaded in this process (no communicator is created, nothing blocks): what the ranks tell each other
// BEFORE anybody enters ncclCommInitRank, a collective without a timeout
int s3_comm_available(void) { return rccl() != nullptr ? 1 : 0; }

int s3_comm_unique_id(void *id_out, size_t bytes) {
    Rccl *R = rccl();
    S3_REQUIRE(R != nullptr, "s3_comm: librccl.so.1 could not be loaded");
    S3_REQUIRE(id_out != nullptr && bytes >= sizeof(ncclUniqueId), "s3_comm_unique_id: the id needs %zu bytes", sizeof(ncclUniqueId));
    ncclUniqueId id;
    S3_RCCL(R->GetUniqueId(&id));
    std::memset(id_out, 0, bytes);
    std::memcpy(id_out, &id, sizeof(id));
    return S3_OK;
}

int s3_comm_init(const void *id, size_t bytes, int rank, int world, s3_comm **out) {
    S3_REQUIRE(out != nullptr, "s3_comm_init: null output");
    *out = nullptr;
    Rccl *R = rccl();
    S3_REQUIRE(R != nullptr, "s3_comm: librccl.so.1 could not be loaded");
    S3_REQUIRE(id != nullptr && bytes >= sizeof(ncclUniqueId) && world >= 1 && rank >= 0 && rank < world,
               "s3_comm_init: bad arguments (rank %d of %d)", rank, world);
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    s3_comm *c = new s3_comm();
    c->rank = rank;
    c->world = world;
    const ncclResult_t r = R->CommInitRank(&c->comm, world, uid, rank);
    if (r != ncclSuccess) {
        s3::set_error("ncclCommInitRank failed: %s", R->GetErrorString(r));
        delete c;
        return S3_EHIP;
    }
    *out = c;
    return S3_OK;
}

void s3_comm_destroy(s3_comm *c) {
    if (!c) return;
    Rccl *R = rccl();
    if (R && c->comm) (void)R->CommDestroy(c->comm);
    delete c;
}

int s3_comm_rank(const s3_comm *c, int *rank, int *world) {
    S3_REQUIRE(c != nullptr, "s3_comm_rank: null communicator");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    return S3_OK;
}

// n_arrays in-place all-gathers issued as ONE group: array a holds world * bytes_per_rank[a] bytes, rank r's part
// starts at r * bytes_per_rank[a] and is sent from there (the exchange of one refine batch: metric, gain)
int s3_comm_allgather_inplace(s3_comm *c, void *const *d_arrays, const size_t *bytes_per_rank, int n_arrays, s3_stream stream) {
    S3_REQUIRE(c != nullptr && d_arrays != nullptr && bytes_per_rank != nullptr && n_arrays >= 1, "s3_comm_allgather_inplace: bad arguments");
    Rccl *R = rccl();
    S3_REQUIRE(R != nullptr, "s3_comm: librccl.so.1 could not be loaded");
    hipStream_t st = s3::as_stream(stream);
    S3_RCCL(R->GroupStart());
    for (int a = 0; a < n_arrays; ++a) {
        if (bytes_per_rank[a] == 0) continue;
        char *base = static_cast<char *>(d_arrays[a]);
        const ncclResult_t r = R->AllGather(base + (size_t)c->rank * bytes_per_rank[a], base, bytes_per_rank[a], ncclUint8, c->comm, st);
        if (r != ncclSuccess) {
            (void)R->GroupEnd();
            s3::set_error("ncclAllGather failed: %s", R->GetErrorString(r));
            return S3_EHIP;
        }
    }
    S3_RCCL(R->GroupEnd());
    return S3_OK;
}

// Every rank's block to ONE rank: rank r sends bytes_per_rank[r] bytes from d_send; `root` receives them one after the other
// (rank order) into d_recv -- its own block is copied on the device.  One group of point-to-point calls: the bytes cross
// xGMI once, to the rank that writes the file, where an all-gather would deliver all of them to every rank.
int s3_comm_gather_to_root(s3_comm *c, const void *d_send, void *d_recv, const size_t *bytes_per_rank, int root, s3_stream stream) {
    S3_REQUIRE(c != nullptr && bytes_per_rank != nullptr && root >= 0 && root < c->world, "s3_comm_gather_to_root: bad arguments");
    Rccl *R = rccl();
    S3_REQUIRE(R != nullptr, "s3_comm: librccl.so.1 could not be loaded");
    const size_t mine = bytes_per_rank[c->rank];
    S3_REQUIRE(mine == 0 || d_send != nullptr, "s3_comm_gather_to_root: null send buffer");
    S3_REQUIRE(c->rank != root || d_recv != nullptr, "s3_comm_gather_to_root: the root needs a receive buffer");
    hipStream_t st = s3::as_stream(stream);
    if (c->rank != root) {
        if (mine) S3_RCCL(R->Send(d_send, mine, ncclUint8, root, c->comm, st));
        return S3_OK;
    }
    S3_RCCL(R->GroupStart());
    size_t off = 0;
    for (int r = 0; r < c->world; ++r) {
        char *dst = static_cast<char *>(d_recv) + off;
        off += bytes_per_rank[r];
        if (bytes_per_rank[r] == 0) continue;
        if (r == root) {
            if (dst != d_send) {
                const hipError_t e = hipMemcpyAsync(dst, d_send, mine, hipMemcpyDeviceToDevice, st);
                if (e != hipSuccess) {
                    (void)R->GroupEnd();
                    s3::set_error("s3_comm_gather_to_root: device copy failed: %s", hipGetErrorString(e));
                    return S3_EHIP;
                }
            }
            continue;
        }
        const ncclResult_t rr = R->Recv(dst, bytes_per_rank[r], ncclUint8, r, c->comm, st);
        if (rr != ncclSuccess) {
            (void)R->GroupEnd();
            s3::set_error("ncclRecv failed: %s", R->GetErrorString(rr));
            return S3_EHIP;
        }
    }
    S3_RCCL(R->GroupEnd());
    return S3_OK;
}

// in-place all-reduce of n doubles; op 0 = sum, 1 = max (wall-clock of the slowest rank)
int s3_comm_allreduce_f64(s3_comm *c, double *d_buf, int64_t n, int op, s3_stream stream) {
    S3_REQUIRE(c != nullptr && d_buf != nullptr && n >= 1 && (op == 0 || op == 1), "s3_comm_allreduce_f64: bad arguments");
    Rccl *R = rccl();
    S3_REQUIRE(R != nullptr, "s3_comm: librccl.so.1 could not be loaded");
    S3_RCCL(R->AllReduce(d_buf, d_buf, (size_t)n, ncclDouble, op == 0 ? ncclSum : ncclMax, c->comm, s3::as_stream(stream)));
    return S3_OK;
}

}  // extern "C"
