// The multi-GPU smoothing step inside the library: sample pass -> ONE all-reduce of the (T,P) f64
// statistics (RCCL over xGMI) -> solve, enqueued on the caller's stream and, optionally, captured once into a
// HIP graph that is replayed with one call per step.
//
// This replaces the reference's ZeroMQ PUSH/PULL fan-out of (x_t, u_t) tasks to 18-30 worker processes
// (zmq_parallel_cmp/array_io.py:6-26, irs_lqr/irs_lqr_quasistatic.py:245-263): samples are sharded over the
// ranks (one process per GPU), nothing but the small sufficient statistics crosses the links, every rank runs
// the tiny solve redundantly.  RCCL is bound at run time (dlopen: the process usually has it loaded already
// through torch.distributed's "nccl" backend, and that copy is then the one used); a host without RCCL can
// still load this library, only these entry points fail.
#include <dlfcn.h>

#include <cstdlib>

#include "irs_common.hpp"

namespace {

// the few RCCL declarations used (rccl.h: ncclUniqueId is 128 opaque bytes, ncclFloat64 = 8, ncclSum = 0)
struct UniqueId { char internal[128]; };
typedef void* Comm;
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(Comm*, int, UniqueId, int);
typedef int (*CommDestroyFn)(Comm);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef const char* (*ErrStrFn)(int);
constexpr int kNcclFloat64 = 8, kNcclSum = 0;

struct Rccl {
    void* h = nullptr;
    GetUniqueIdFn get_id = nullptr;
    CommInitRankFn init = nullptr;
    CommDestroyFn destroy = nullptr;
    AllReduceFn all_reduce = nullptr;
    ErrStrFn err = nullptr;
    bool ok = false;
};

Rccl& rccl() {
    static Rccl r;
    static bool tried = false;
    if (!tried) {
        tried = true;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.h) break;
        }
        if (r.h) {
            r.get_id = (GetUniqueIdFn)dlsym(r.h, "ncclGetUniqueId");
            r.init = (CommInitRankFn)dlsym(r.h, "ncclCommInitRank");
            r.destroy = (CommDestroyFn)dlsym(r.h, "ncclCommDestroy");
            r.all_reduce = (AllReduceFn)dlsym(r.h, "ncclAllReduce");
            r.err = (ErrStrFn)dlsym(r.h, "ncclGetErrorString");
            r.ok = r.get_id && r.init && r.destroy && r.all_reduce;
        }
    }
    return r;
}

int need_rccl(const char* who) {
    if (!rccl().ok) {
        irs_set_error("%s: RCCL (librccl.so) could not be loaded: %s", who, dlerror() ? dlerror() : "symbols missing");
        return IRS_ERR_UNSUPPORTED;
    }
    return IRS_OK;
}

int nccl_fail(const char* who, int rc) {
    irs_set_error("%s: RCCL error %d (%s)", who, rc, rccl().err ? rccl().err(rc) : "?");
    return IRS_ERR_HIP;
}

struct StepGraph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};

// ---- the exchange WITHOUT a collective library: peers' memory mapped by IPC handle, one small launch ----------
// An 8-rank RCCL all-reduce of 17 KB is latency (tens of us: a ring of point-to-point xGMI hops, each a
// flag-polled hand-off) under a sample pass of ~35 us.  The statistics are tiny, every rank needs all of them and
// xGMI is all-to-all: each rank PUBLISHES its (T,P) block in its own exchange region and READS the other ranks'
// blocks straight out of theirs -- one hop, every link used at once, and the sum is taken in rank order, so every
// rank holds the same bits.  Opt-in (bench.py --collective peer): RCCL stays the default until a multi-GPU node has
// validated the memory model below; what a one-GPU box can test (mapping, flags, step logic, buffer reuse, two
// ranks on one device) is tested.
//
// Region of a rank (device memory, exported by hipIpcGetMemHandle):  [step flag, 128 bytes][slot 0][slot 1]
// Memory model: every access to a region -- the publishing stores, the flag, the readers' loads -- is a
// SYSTEM-scope atomic (sc0 sc1 on gfx950: written through / read past the non-coherent caches); the flag is stored
// with release after a system-scope fence that follows a workgroup barrier (all of the block's publishing stores
// precede it), and read with acquire.  Slot reuse: the flag of step k+1 is stored by the peer's launch k+1, which is
// stream-ordered after its launch k has finished reading; a rank that has seen every flag >= k+1 may therefore
// overwrite slot (k+2) & 1 = k & 1.  A peer that never arrives ends the wait after kPeerSpinTicks of the constant
// 100 MHz clock: the statistics are poisoned (NaN), which the solve reports through `info` like any other
// non-finite statistic -- a broken job fails, it does not hang the GPU.
constexpr int kMaxPeers = 16, kPeerHdr = 16 /* doubles */, kPeerBlock = 1024;
constexpr unsigned long long kPeerSpinTicksDefault = 200000000ull;      // 2 s of s_memrealtime

struct PeerDev {
    double* region[kMaxPeers];
    unsigned long long* step;           // this rank's step counter (device; advanced by the launch itself: graph replays)
    unsigned long long spin_ticks;
    size_t count;
    int nranks, rank;
};

struct PeerX {
    PeerDev d;
    void* mapped[kMaxPeers];            // what hipIpcOpenMemHandle returned (nullptr for the own region)
    unsigned long long* stats;          // device: [0] launches, [1] timeouts
};

__global__ __launch_bounds__(kPeerBlock) void peer_exchange_kernel(PeerDev p, double* sums, unsigned long long* stats) {
    __shared__ unsigned long long s_step;
    __shared__ int s_bad;
    const int tid = threadIdx.x;
    if (tid == 0) {
        s_step = *p.step + 1ull;
        *p.step = s_step;
        s_bad = 0;
    }
    __syncthreads();
    const unsigned long long step = s_step;
    const size_t slot = kPeerHdr + (size_t)(step & 1ull) * p.count;
    double* mine = p.region[p.rank] + slot;
    for (size_t i = tid; i < p.count; i += kPeerBlock)
        __hip_atomic_store(&mine[i], sums[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __atomic_thread_fence(__ATOMIC_RELEASE);            // this thread's publishing stores, system scope (HIP default)
    __syncthreads();
    if (tid == 0)
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(p.region[p.rank]), step, __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    if (tid < p.nranks) {
        const unsigned long long* flag = reinterpret_cast<const unsigned long long*>(p.region[tid]);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < step) {
            if (__builtin_amdgcn_s_memrealtime() - t0 > p.spin_ticks) {
                s_bad = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    if (s_bad) {
        for (size_t i = tid; i < p.count; i += kPeerBlock) sums[i] = __builtin_nan("");
        if (tid == 0 && stats) { stats[0] += 1ull; stats[1] += 1ull; }
        return;
    }
    for (size_t i = tid; i < p.count; i += kPeerBlock) {
        double tot = 0.0;
        for (int r = 0; r < p.nranks; ++r)
            tot += __hip_atomic_load(p.region[r] + slot + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        sums[i] = tot;
    }
    if (tid == 0 && stats) stats[0] += 1ull;
}

int peer_enqueue(PeerX* px, double* sums, size_t count, hipStream_t st) {
    if (count != px->d.count) {
        irs_set_error("peer exchange made for %zu doubles, called with %zu", px->d.count, count);
        return IRS_ERR_INVALID_ARG;
    }
    hipLaunchKernelGGL(peer_exchange_kernel, dim3(1), dim3(kPeerBlock), 0, st, px->d, sums, px->stats);
    return IRS_OK;
}

// the three enqueues of one multi-GPU smoothing step
int enqueue_step(const irs_smooth_call* c, Comm comm, hipStream_t st, PeerX* peer = nullptr) {
    IRS_CHECK_ARG(c != nullptr && c->sums != nullptr && c->At && c->Bt && c->ct && c->info, "the call needs sums and At/Bt/ct/info");
    irs_smooth_call acc = *c;
    acc.At = nullptr; acc.Bt = nullptr; acc.ct = nullptr; acc.info = nullptr;      // accumulate only
    int rc = irs_smooth_run(&acc, st);
    if (rc != IRS_OK) return rc;
    const size_t count = (size_t)c->T * (size_t)irs_sums_len(c->model, c->mode);
    if (peer != nullptr) {
        rc = peer_enqueue(peer, c->sums, count, st);
        if (rc != IRS_OK) return rc;
    } else if (comm != nullptr) {
        const int nrc = rccl().all_reduce(c->sums, c->sums, count, kNcclFloat64, kNcclSum, comm, st);
        if (nrc != 0) return nccl_fail("irs_smooth_step_collective", nrc);
    }
    return irs_smooth_finalize_ws(c->model, c->params, c->n_params, c->mode, c->T, c->n_total, c->x_trj, c->u_trj,
                                  c->sums, c->At, c->Bt, c->ct, c->info, c->workspace, c->workspace_bytes, st);
}

}  // namespace

extern "C" {

int irs_comm_available(void) { return rccl().ok ? 1 : 0; }

int irs_comm_unique_id(void* id128) {
    IRS_CHECK_ARG(id128 != nullptr, "null id buffer");
    int rc = need_rccl("irs_comm_unique_id");
    if (rc != IRS_OK) return rc;
    const int nrc = rccl().get_id(static_cast<UniqueId*>(id128));
    return nrc == 0 ? IRS_OK : nccl_fail("irs_comm_unique_id", nrc);
}

int irs_comm_create(const void* id128, int nranks, int rank, void** comm) {
    IRS_CHECK_ARG(id128 != nullptr && comm != nullptr && nranks > 0 && rank >= 0 && rank < nranks, "bad argument");
    int rc = need_rccl("irs_comm_create");
    if (rc != IRS_OK) return rc;
    UniqueId id;
    memcpy(&id, id128, sizeof(id));
    Comm c = nullptr;
    const int nrc = rccl().init(&c, nranks, id, rank);          // collective: every rank calls it, current device
    if (nrc != 0) return nccl_fail("irs_comm_create", nrc);
    *comm = c;
    return IRS_OK;
}

int irs_comm_destroy(void* comm) {
    if (comm == nullptr) return IRS_OK;
    int rc = need_rccl("irs_comm_destroy");
    if (rc != IRS_OK) return rc;
    const int nrc = rccl().destroy(comm);
    return nrc == 0 ? IRS_OK : nccl_fail("irs_comm_destroy", nrc);
}

int irs_allreduce_sums(void* comm, double* sums, size_t count, void* stream) {
    IRS_CHECK_ARG(comm != nullptr && sums != nullptr && count > 0, "bad argument");
    int rc = need_rccl("irs_allreduce_sums");
    if (rc != IRS_OK) return rc;
    const int nrc = rccl().all_reduce(sums, sums, count, kNcclFloat64, kNcclSum, comm, static_cast<hipStream_t>(stream));
    return nrc == 0 ? IRS_OK : nccl_fail("irs_allreduce_sums", nrc);
}

int irs_smooth_step_collective(const irs_smooth_call* call, void* comm, void* stream) {
    if (comm != nullptr) {
        int rc = need_rccl("irs_smooth_step_collective");
        if (rc != IRS_OK) return rc;
    }
    return enqueue_step(call, comm, static_cast<hipStream_t>(stream));
}

static int step_graph_create(const irs_smooth_call* call, void* comm, PeerX* peer, void* stream, void** graph_exec) {
    IRS_CHECK_ARG(call != nullptr && graph_exec != nullptr && stream != nullptr, "needs a call, a non-default stream and an out pointer");
    if (comm != nullptr) {
        int rc = need_rccl("irs_step_graph_create");
        if (rc != IRS_OK) return rc;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    StepGraph* g = new StepGraph();
    hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) {
        delete g;
        irs_set_error("irs_step_graph_create: hipStreamBeginCapture: %s", hipGetErrorString(e));
        return IRS_ERR_HIP;
    }
    const int rc = enqueue_step(call, comm, st, peer);
    e = hipStreamEndCapture(st, &g->graph);
    if (rc != IRS_OK || e != hipSuccess || g->graph == nullptr) {
        if (g->graph) (void)hipGraphDestroy(g->graph);
        delete g;
        if (rc == IRS_OK) irs_set_error("irs_step_graph_create: hipStreamEndCapture: %s", hipGetErrorString(e));
        return rc != IRS_OK ? rc : IRS_ERR_HIP;
    }
    e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGraphDestroy(g->graph);
        delete g;
        irs_set_error("irs_step_graph_create: hipGraphInstantiate: %s", hipGetErrorString(e));
        return IRS_ERR_HIP;
    }
    *graph_exec = g;
    return IRS_OK;
}

int irs_step_graph_create(const irs_smooth_call* call, void* comm, void* stream, void** graph_exec) {
    return step_graph_create(call, comm, nullptr, stream, graph_exec);
}

int irs_step_graph_create_peer(const irs_smooth_call* call, void* peer, void* stream, void** graph_exec) {
    IRS_CHECK_ARG(peer != nullptr, "null peer exchange");
    return step_graph_create(call, nullptr, static_cast<PeerX*>(peer), stream, graph_exec);
}

// ---- peer exchange: set-up ------------------------------------------------------------------------------------
int irs_peer_alloc(size_t count, void** region, void* handle64) {
    IRS_CHECK_ARG(count > 0 && region != nullptr && handle64 != nullptr, "bad argument");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the handle travels as 64 bytes");
    const size_t bytes = (kPeerHdr + 2 * count) * sizeof(double);
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e == hipSuccess) e = hipMemset(p, 0, bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    hipIpcMemHandle_t h;
    if (e == hipSuccess) e = hipIpcGetMemHandle(&h, p);
    if (e != hipSuccess) {
        if (p) (void)hipFree(p);
        irs_set_error("irs_peer_alloc: %s", hipGetErrorString(e));
        return IRS_ERR_HIP;
    }
    memcpy(handle64, &h, sizeof(h));
    *region = p;
    return IRS_OK;
}

int irs_peer_create(int nranks, int rank, void* region, size_t count, const void* handles, void** peer) {
    IRS_CHECK_ARG(nranks > 0 && nranks <= kMaxPeers && rank >= 0 && rank < nranks && region != nullptr && count > 0 &&
                  handles != nullptr && peer != nullptr, "bad argument (at most 16 ranks)");
    PeerX* px = new PeerX();
    memset(px, 0, sizeof(*px));
    px->d.nranks = nranks;
    px->d.rank = rank;
    px->d.count = count;
    px->d.spin_ticks = kPeerSpinTicksDefault;
    if (const char* ms = getenv("IRS_PEER_TIMEOUT_MS")) {
        const long v = atol(ms);
        if (v > 0) px->d.spin_ticks = (unsigned long long)v * 100000ull;
    }
    auto fail = [&](const char* what, hipError_t e) {
        irs_set_error("irs_peer_create: %s: %s", what, hipGetErrorString(e));
        for (int r = 0; r < nranks; ++r)
            if (px->mapped[r]) (void)hipIpcCloseMemHandle(px->mapped[r]);
        if (px->d.step) (void)hipFree(px->d.step);
        delete px;
        return IRS_ERR_HIP;
    };
    for (int r = 0; r < nranks; ++r) {
        if (r == rank) {
            px->d.region[r] = static_cast<double*>(region);
            continue;
        }
        hipIpcMemHandle_t h;
        memcpy(&h, static_cast<const char*>(handles) + (size_t)r * sizeof(h), sizeof(h));
        void* m = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&m, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) return fail("hipIpcOpenMemHandle", e);
        px->mapped[r] = m;
        px->d.region[r] = static_cast<double*>(m);
    }
    void* ctr = nullptr;
    hipError_t e = hipMalloc(&ctr, 4 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(ctr, 0, 4 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) return fail("hipMalloc", e);
    px->d.step = static_cast<unsigned long long*>(ctr);
    px->stats = px->d.step + 1;
    *peer = px;
    return IRS_OK;
}

int irs_peer_destroy(void* peer, void* region) {
    hipError_t e = hipDeviceSynchronize();
    if (peer != nullptr) {
        PeerX* px = static_cast<PeerX*>(peer);
        for (int r = 0; r < px->d.nranks; ++r)
            if (px->mapped[r]) (void)hipIpcCloseMemHandle(px->mapped[r]);
        if (px->d.step) (void)hipFree(px->d.step);
        delete px;
    }
    if (region != nullptr && e == hipSuccess) e = hipFree(region);
    if (e != hipSuccess) {
        irs_set_error("irs_peer_destroy: %s", hipGetErrorString(e));
        return IRS_ERR_HIP;
    }
    return IRS_OK;
}

int irs_peer_status(void* peer, unsigned long long* launches, unsigned long long* timeouts) {
    IRS_CHECK_ARG(peer != nullptr, "null peer exchange");
    unsigned long long h[2] = {0, 0};
    const hipError_t e = hipMemcpy(h, static_cast<PeerX*>(peer)->stats, sizeof(h), hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
        irs_set_error("irs_peer_status: %s", hipGetErrorString(e));
        return IRS_ERR_HIP;
    }
    if (launches) *launches = h[0];
    if (timeouts) *timeouts = h[1];
    return IRS_OK;
}

int irs_peer_allreduce_sums(void* peer, double* sums, size_t count, void* stream) {
    IRS_CHECK_ARG(peer != nullptr && sums != nullptr && count > 0, "bad argument");
    return peer_enqueue(static_cast<PeerX*>(peer), sums, count, static_cast<hipStream_t>(stream));
}

int irs_smooth_step_peer(const irs_smooth_call* call, void* peer, void* stream) {
    IRS_CHECK_ARG(peer != nullptr, "null peer exchange");
    return enqueue_step(call, nullptr, static_cast<hipStream_t>(stream), static_cast<PeerX*>(peer));
}

int irs_step_graph_launch(void* graph_exec, void* stream) {
    IRS_CHECK_ARG(graph_exec != nullptr, "null graph");
    const hipError_t e = hipGraphLaunch(static_cast<StepGraph*>(graph_exec)->exec, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) {
        irs_set_error("irs_step_graph_launch: %s", hipGetErrorString(e));
        return IRS_ERR_HIP;
    }
    return IRS_OK;
}

int irs_step_graph_destroy(void* graph_exec) {
    if (graph_exec == nullptr) return IRS_OK;
    StepGraph* g = static_cast<StepGraph*>(graph_exec);
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
    return IRS_OK;
}

}  // extern "C"
