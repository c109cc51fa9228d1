"""Sample sharding over the GPUs of one node.

The reference's only parallelism is a ZeroMQ PUSH/PULL fan-out of (x_t,u_t) tasks to
18-30 worker processes, each returning a finished (A_t,B_t) (zmq_parallel_cmp/
array_io.py:6-26, irs_lqr/irs_lqr_quasistatic.py:245-263).  Here the N i.i.d. samples
of every timestep are split into contiguous shards, one per rank (= one per GPU);
each rank reduces its shard to the per-timestep sufficient statistics `sums (T,P)`
and ONE all-reduce (RCCL over xGMI; `nccl` backend on ROCm) of that small f64 buffer
replaces all the ZMQ traffic.  Every rank then runs the tiny solve + Riccati +
rollout redundantly (cheaper than a broadcast).  No collective touches the samples.
"""
import torch
import torch.distributed as dist


def rank_world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(N, rank, world):
    """Contiguous, near-equal split of range(N): rank r owns [lo, hi)."""
    base, rem = divmod(int(N), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def all_reduce_sums(sums, group=None):
    """In-place SUM all-reduce of the (T,P) statistics; no-op for a single rank.
    `nccl` (= RCCL on ROCm) reduces the device tensor in place over xGMI; with the `gloo`
    backend (CPU rehearsals of the N>1 path) a device tensor is staged through the host."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if sums.is_cuda and dist.get_backend(group) == "gloo":
            host = sums.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            sums.copy_(host)
        else:
            dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    return sums


def capture_step(fn, warmup=3):
    """Captures `fn` -- a fixed sequence of launches on torch's current stream: C-ABI kernels, an RCCL
    all-reduce, ... -- into a HIP graph and returns the replay callable: one host call per step instead
    of one per launch (the multi-GPU smoothing step is three launches of ~70 us together, which a Python
    host cannot issue fast enough one by one).  `fn` must read the stream from
    torch.cuda.current_stream() at call time and must not allocate.  The capture runs on every rank at the
    same point (the warm-up calls contain the collective)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warmup):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        fn()
    torch.cuda.synchronize()
    return graph.replay


class DirectComm:
    """An RCCL communicator owned by the HIP library (include/irs_hip.h, irs_comm_*): the unique id is made on
    rank 0 and handed to the other ranks through torch.distributed (whatever its backend: the id is 128 host
    bytes); every rank then joins with its current device.  With it the whole multi-GPU smoothing step --
    sample pass, all-reduce of the (T,P) sums, solve -- is enqueued, or captured into one HIP graph, INSIDE the
    library (`CollectiveStep`), without torch's collectives in the data path."""

    def __init__(self, group=None):
        import ctypes
        from . import _lib
        self.lib = _lib.load()
        self.rank, self.world = rank_world()
        ident = (ctypes.c_char * 128)()
        if self.rank == 0:
            _lib.check(self.lib.irs_comm_unique_id(ctypes.cast(ident, ctypes.c_void_p)), "irs_comm_unique_id")
        if self.world > 1:
            t = torch.frombuffer(bytearray(bytes(ident)), dtype=torch.uint8).clone()
            if dist.get_backend(group) == "nccl":
                t = t.cuda()
            dist.broadcast(t, src=0, group=group)
            ident = (ctypes.c_char * 128).from_buffer_copy(bytes(t.cpu().numpy().tobytes()))
        self.handle = ctypes.c_void_p()
        _lib.check(self.lib.irs_comm_create(ctypes.cast(ident, ctypes.c_void_p), self.world, self.rank,
                                            ctypes.byref(self.handle)), "irs_comm_create")

    def all_reduce_sums(self, sums):
        from . import _lib
        _lib.check(self.lib.irs_allreduce_sums(self.handle, sums.data_ptr(), sums.numel(),
                                               torch.cuda.current_stream().cuda_stream), "irs_allreduce_sums")
        return sums

    def destroy(self):
        if getattr(self, "handle", None) is not None and self.handle.value:
            self.lib.irs_comm_destroy(self.handle)
            self.handle = None


class PeerExchange:
    """The all-reduce of the (T,P) sums WITHOUT a collective library (include/irs_hip.h, irs_peer_*): every rank
    publishes its block in an exchange region of its own device memory and reads the other ranks' regions, mapped by
    IPC handle, directly; the blocks are added in rank order, so every rank holds the same bits.  The 64-byte
    handles travel through torch.distributed (any backend).  Opt-in: RCCL (`DirectComm`) stays the default until a
    multi-GPU node has validated the exchange's memory model."""

    def __init__(self, count, group=None):
        import ctypes
        from . import _lib
        self.lib = _lib.load()
        self.rank, self.world = rank_world()
        self.count = int(count)
        self.region = ctypes.c_void_p()
        mine = (ctypes.c_char * 64)()
        _lib.check(self.lib.irs_peer_alloc(self.count, ctypes.byref(self.region), ctypes.cast(mine, ctypes.c_void_p)),
                   "irs_peer_alloc")
        blob = bytes(mine)
        if self.world > 1:
            t = torch.frombuffer(bytearray(blob), dtype=torch.uint8).clone()
            on_dev = dist.get_backend(group) == "nccl"
            if on_dev:
                t = t.cuda()
            parts = [torch.empty_like(t) for _ in range(self.world)]
            dist.all_gather(parts, t, group=group)
            blob = b"".join(bytes(p_.cpu().numpy().tobytes()) for p_ in parts)
        self._handles = (ctypes.c_char * (64 * self.world)).from_buffer_copy(blob)
        self.handle = ctypes.c_void_p()
        _lib.check(self.lib.irs_peer_create(self.world, self.rank, self.region, self.count,
                                            ctypes.cast(self._handles, ctypes.c_void_p), ctypes.byref(self.handle)),
                   "irs_peer_create")
        if self.world > 1:
            dist.barrier(group=group)           # every rank has mapped every region before the first launch

    def all_reduce_sums(self, sums):
        from . import _lib
        _lib.check(self.lib.irs_peer_allreduce_sums(self.handle, sums.data_ptr(), sums.numel(),
                                                    torch.cuda.current_stream().cuda_stream), "irs_peer_allreduce_sums")
        return sums

    def status(self):
        """(launches, timeouts) so far; synchronises."""
        import ctypes
        from . import _lib
        a, b = ctypes.c_ulonglong(), ctypes.c_ulonglong()
        _lib.check(self.lib.irs_peer_status(self.handle, ctypes.byref(a), ctypes.byref(b)), "irs_peer_status")
        return int(a.value), int(b.value)

    def destroy(self, group=None):
        if getattr(self, "handle", None) is not None and self.handle.value:
            torch.cuda.synchronize()
            if self.world > 1:
                dist.barrier(group=group)       # nobody unmaps while a peer may still read
            self.lib.irs_peer_destroy(self.handle, self.region)
            self.handle = None


class CollectiveStep:
    """One multi-GPU smoothing step of a `SmoothPlan` built with fuse=True outputs AND sums (accumulate ->
    all-reduce -> solve), issued by the library: `run()` enqueues the three launches, `capture()` records them
    once into a HIP graph and `run()` then replays it with a single call."""

    def __init__(self, plan, comm=None):
        import ctypes
        from . import _lib
        self.plan, self.comm, self.lib = plan, comm, _lib.load()
        self._ref = ctypes.byref(plan.call)
        self._graph = ctypes.c_void_p()
        self._check = _lib.check
        assert plan.out is not None, "build the SmoothPlan with fuse=True: the step needs At/Bt/ct/info"

    def _comm(self):
        return self.comm.handle if self.comm is not None else None

    def _enqueue(self, st):
        if isinstance(self.comm, PeerExchange):
            self._check(self.lib.irs_smooth_step_peer(self._ref, self.comm.handle, st), "irs_smooth_step_peer")
        else:
            self._check(self.lib.irs_smooth_step_collective(self._ref, self._comm(), st), "irs_smooth_step_collective")

    def capture(self):
        import ctypes
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):              # warm-up on the capture stream (RCCL sets its channels up lazily)
                self._enqueue(side.cuda_stream)
            side.synchronize()
            if isinstance(self.comm, PeerExchange):
                self._check(self.lib.irs_step_graph_create_peer(self._ref, self.comm.handle, side.cuda_stream,
                                                                ctypes.byref(self._graph)), "irs_step_graph_create_peer")
            else:
                self._check(self.lib.irs_step_graph_create(self._ref, self._comm(), side.cuda_stream,
                                                           ctypes.byref(self._graph)), "irs_step_graph_create")
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        return self

    def run(self):
        st = torch.cuda.current_stream().cuda_stream
        if self._graph.value:
            self._check(self.lib.irs_step_graph_launch(self._graph, st), "irs_step_graph_launch")
        else:
            self._enqueue(st)
        return self.plan.out

    def destroy(self):
        if self._graph.value:
            self.lib.irs_step_graph_destroy(self._graph)
            self._graph.value = None
