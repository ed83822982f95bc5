"""CPU, world_size 2 over gloo: the N>1 path of bench.py — contiguous image sharding, barrier +
max-over-ranks timing, whole-job aggregate — has no data-path collective.  The GPU
kernels cannot run here; the harness logic is exercised with the CPU oracle standing in for the
per-rank forward, and shard equivalence (B images on 1 rank == the same images split over 2 ranks,
bitwise) is checked on it."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def shard(n_items, rank, world):
    """contiguous split used for image sharding: rank r gets [r*n/world, (r+1)*n/world)"""
    lo = rank * n_items // world
    hi = (rank + 1) * n_items // world
    return lo, hi


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import c_oracle as co
    erp = np.random.default_rng(5).random((4, 1, 32, 64), dtype=np.float32)        # the global batch
    lo, hi = shard(4, rank, world)
    mine, _, _, _ = co.equi2pers(erp[lo:hi], 80, 4, 8)
    back = co.pers2equi(mine, 80, 4, 8, (32, 64))
    # timing protocol of bench.py: barrier, local time, MAX over ranks
    dist.barrier()
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    # optional result gather (teardown only, not on the data path)
    outs = [torch.zeros(2, 1, 32, 64) for _ in range(world)]
    dist.all_gather(outs, torch.from_numpy(back))
    if rank == 0:
        q.put((float(t.item()), torch.cat(outs).numpy()))
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_rank():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    tmax, gathered = q.get(timeout=240)
    for p in procs:
        p.join(60); assert p.exitcode == 0
    assert abs(tmax - 0.2) < 1e-12                                   # MAX over ranks
    sys.path.insert(0, ROOT)
    from oracle import c_oracle as co
    erp = np.random.default_rng(5).random((4, 1, 32, 64), dtype=np.float32)
    full, _, _, _ = co.equi2pers(erp, 80, 4, 8)
    ref = co.pers2equi(full, 80, 4, 8, (32, 64))
    assert np.array_equal(gathered, ref)                              # bitwise shard equivalence


def test_shard_bounds_cover_batch_exactly():
    for n in (1, 7, 8, 64):
        for w in (1, 2, 4, 8):
            parts = [shard(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
