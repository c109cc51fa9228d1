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

// the three enqueues of one multi-GPU smoothing step
int enqueue_step(const irs_smooth_call* c, Comm comm, hipStream_t st) {
    IRS_CHECK_ARG(c != nullptr && c->sums != nullptr && c->At && c->Bt && c->ct && c->info, "the call needs sums and At/Bt/ct/info");
    irs_smooth_call acc = *c;
    acc.At = nullptr; acc.Bt = nullptr; acc.ct = nullptr; acc.info = nullptr;      // accumulate only
    int rc = irs_smooth_run(&acc, st);
    if (rc != IRS_OK) return rc;
    const size_t count = (size_t)c->T * (size_t)irs_sums_len(c->model, c->mode);
    if (comm != nullptr) {
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

int irs_step_graph_create(const irs_smooth_call* call, void* comm, void* stream, void** graph_exec) {
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
    const int rc = enqueue_step(call, comm, st);
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
