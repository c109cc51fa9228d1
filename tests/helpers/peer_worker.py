"""One rank of the peer-exchange functional test (tests/test_gpu_parity.py): launched by torch.distributed.run with
two (or more) ranks that ALL use GPU 0; gloo carries the 64-byte IPC handles.  Every rank fills its statistics with a
rank- and step-dependent pattern, runs the exchange launch back to back (slot reuse, no host synchronisation in
between) and checks the rank-ordered total it gets against the closed form."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from irs_mpc_amd.distributed import PeerExchange  # noqa: E402


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    count, steps = 2125, 300
    px = PeerExchange(count)
    base = torch.arange(count, dtype=torch.float64, device="cuda")
    bufs = [torch.empty(count, dtype=torch.float64, device="cuda") for _ in range(steps)]
    for k in range(steps):                      # pattern(r, k, i) = (r + 1) * (i + 0.25 k) + 1e-3 r k
        bufs[k].copy_((rank + 1) * (base + 0.25 * k) + 1e-3 * rank * k)
    torch.cuda.synchronize()
    dist.barrier()
    for k in range(steps):
        px.all_reduce_sums(bufs[k])
    torch.cuda.synchronize()
    launches, timeouts = px.status()
    assert (launches, timeouts) == (steps, 0), (launches, timeouts)
    i = np.arange(count, dtype=np.float64)
    for k in range(steps):
        want = np.zeros(count)
        for r in range(world):                  # the same order as the kernel: rank 0 first
            want = want + ((r + 1) * (i + 0.25 * k) + 1e-3 * r * k)
        got = bufs[k].cpu().numpy()
        assert np.array_equal(got, want), (rank, k, float(np.abs(got - want).max()))
    # every rank holds the same bits
    mine = torch.stack(bufs).cpu()
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    assert all(torch.equal(parts[0], p) for p in parts)
    # a peer that does not arrive: rank 0 issues one exchange more than the others.  Its wait must END (bounded by
    # IRS_PEER_TIMEOUT_MS), poison the statistics and count a timeout -- not hang the GPU
    dist.barrier()
    if "--missing-peer" in sys.argv and world > 1:
        if rank == 0:
            extra = torch.ones(count, dtype=torch.float64, device="cuda")
            px.all_reduce_sums(extra)
            torch.cuda.synchronize()
            launches, timeouts = px.status()
            assert timeouts == 1 and launches == steps + 1, (launches, timeouts)
            assert bool(torch.isnan(extra).all()), "a timed-out exchange must poison the statistics"
            print("PEER_TIMEOUT_OK", flush=True)
        dist.barrier()
    px.destroy()
    dist.barrier()
    if rank == 0:
        print("PEER_EXCHANGE_OK world=%d steps=%d" % (world, steps), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
