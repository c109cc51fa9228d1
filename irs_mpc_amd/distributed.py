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
