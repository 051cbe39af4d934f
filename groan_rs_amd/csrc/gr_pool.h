// gr_pool.h -- the multi-GPU form of System::traj_iter_map_reduce (src/system/parallel.rs:208-481) behind the C ABI.
//
// Two shapes, one sharding rule -- frame f belongs to worker f mod T, exactly the reference's thread sharding
// (parallel.rs:424-448) -- and NO data-path collective: frames are independent, bulk frame data never leaves its GPU.
//
//   gr_pool_*   in-process: one worker thread + one context per listed device (System::clone per thread, parallel.rs:236);
//               the body runs on the worker's thread with the worker's context; a shared error flag is polled every
//               ERROR_FLAG_FREQ = 10 frames (parallel.rs:28,453-475: the first error wins, the others stop at their next
//               check); per-frame results land at their frame's position, so the "gather" is plain host memory
//   gr_comm_*   multi-process, one process per GPU: an RCCL communicator (ncclCommInitRank from a unique id the launcher
//               hands round), ONE ncclAllGather of the per-frame results over xGMI at the end of the run + the host
//               de-interleave out[f] = shard[f mod G][f div G], and the shared error flag as a 1-int ncclAllReduce(MAX).
//               The message is tiny (4-40 B per frame): latency-bound, far from the 7 x ~153 GB/s of the xGMI links.
//
// RCCL is NOT a link-time dependency: its entry points are resolved when the first communicator is created -- from the
// process first (a host that already loaded RCCL, e.g. through torch.distributed, gets THAT copy: two RCCLs in one process
// would each claim the GPU's IPC resources), else from librccl.so.1 of the ROCm installation.
#pragma once
#include <mutex>
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/groan_hip.h"

#define GR_POOL_ERROR_FLAG_FREQ 10   /* parallel.rs:28 */

struct gr_pool {
    std::vector<gr_ctx *> ctx;
    std::vector<int> device;
    std::string err;
};

// ---- RCCL through dlsym (types as in rccl.h: ncclResult_t / ncclDataType_t / ncclRedOp_t are ints, ncclUniqueId is 128 bytes)
namespace grn {
typedef struct ncclComm *comm_t;
struct unique_id { char internal[128]; };
enum { Success = 0, Int32 = 2, Float32 = 7, Max = 2 };
struct Api {
    int (*GetUniqueId)(unique_id *) = nullptr;
    int (*CommInitRank)(comm_t *, int, unique_id, int) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, comm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, comm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string source;
    bool ok = false;
};
// the RCCL library a host wants used instead of the default search (gr_comm_set_library, before the first gr_comm_* call)
// (guarded: a setter on one thread and the first gr_comm_* call on another meet here; `resolved` = the table has been built, a later
// override could not take effect any more and is refused)
struct LibraryChoice { std::mutex mu; std::string path; bool resolved = false; };
inline LibraryChoice &library_choice() { static LibraryChoice c; return c; }
inline bool library_override_set(const char *path) {
    LibraryChoice &c = library_choice();
    std::lock_guard<std::mutex> g(c.mu);
    if (c.resolved) return false;
    c.path = path;
    return true;
}
inline Api load_api() {
    Api a;
    void *h = RTLD_DEFAULT;
    a.source = "already loaded in the process";
    std::string want;
    { LibraryChoice &c = library_choice(); std::lock_guard<std::mutex> g(c.mu); c.resolved = true; want = c.path; }
    if (!want.empty() || !dlsym(h, "ncclCommInitRank")) {
        (void)dlerror();                                    // (clear: the text below must belong to OUR failure)
        if (!want.empty()) { h = dlopen(want.c_str(), RTLD_NOW | RTLD_GLOBAL); a.source = want; }
        else {
            h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            a.source = "librccl.so.1";
        }
        if (!h) {
            const char *e = dlerror();                      // ONE call: it returns the message and clears it
            a.source = std::string("RCCL not found: ") + (e ? e : "(no loader message)");
            return a;
        }
    }
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(h, "ncclAllGather"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(h, "ncclAllReduce"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllGather && a.AllReduce;
    if (!a.ok) a.source += " (symbols missing)";
    return a;
}
// resolved once, by whichever thread asks first (a function-local static: its initialisation is thread-safe, and nobody can see
// a half-filled table)
inline Api &api() {
    static Api a = load_api();
    return a;
}
}  // namespace grn

struct gr_comm {
    grn::comm_t comm = nullptr;
    int device = 0, rank = 0, world = 1;
    hipStream_t stream = nullptr;
    float *send = nullptr, *recv = nullptr; size_t cap = 0;   // device staging, grow-only (floats: send cap, recv cap * world)
    int *flag = nullptr;
    std::string err;
};

// out[f] = shard[f mod G][f div G]: the inverse of the round-robin sharding (order restore after the gather)
inline void gr_deinterleave(const float *gathered /* [G][per][width] */, int world, uint64_t per, size_t width, uint64_t n_total, float *out) {
    for (uint64_t f = 0; f < n_total; ++f)
        memcpy(out + f * width, gathered + ((size_t)(f % (uint64_t)world) * per + f / (uint64_t)world) * width, width * sizeof(float));
}
