"""Image sharding across the GPUs of one node — one process per GPU, `torch.distributed` over RCCL/xGMI.

Replaces the reference's `nn.DataParallel` scatter -> per-forward parameter re-broadcast -> thread-per-GPU ->
gather (`/root/reference/test.py:105-107,198`).  Every panorama is independent end to end in eval mode
(SURVEY.md 8e), so the data path has NO collective: rank r of R processes the contiguous image range
`shard(B, r, R)` with its own replica of the packed weights and geometry tables.  RCCL is used only
  * once at start-up, optionally, to broadcast a checkpoint from rank 0 (`broadcast_state_dict`),
  * at the end, optionally, to collect the per-rank depth maps (`gather_batch`: one all_gather of
    [B/R,1,H,W] — 16.8 MB per GPU at BASELINE cfg 4, a single large message per xGMI link),
  * for the barrier / MAX-over-ranks of the timing protocol (`timed_steps`).

The functions take tensors on whatever device the process group's backend serves ("nccl" = RCCL: the rank's
GPU; "gloo": CPU), which is what lets tests/test_sharding_gloo.py drive exactly this code with world_size 2
on a box without GPUs.
"""
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist


def shard(n_items, rank, world):
    """Contiguous split used for image sharding: rank r gets [r*n/world, (r+1)*n/world).  Covers the
    batch exactly for any n, world (ragged when world does not divide n)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    return rank * n_items // world, (rank + 1) * n_items // world


def env_rank():
    """(rank, local_rank, world) from the launcher's environment (torch.distributed.run); (0, 0, 1) if unset."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def _parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the kernel's cpulist format)"""
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.extend(range(int(a), int(b or a) + 1))
    return cpus


def _fmt_cpus(cpus):
    """[0, 1, 2, 3, 8] -> '0-3,8'"""
    out, run = [], []
    for c in sorted(cpus) + [None]:
        if run and (c is None or c != run[-1] + 1):
            out.append(str(run[0]) if len(run) == 1 else f"{run[0]}-{run[-1]}")
            run = []
        if c is not None:
            run.append(c)
    return ",".join(out)


def _gpu_local_cpus(index):
    """CPUs of the NUMA node the GPU `index` hangs off (sysfs `local_cpulist` of its PCI function), or None when unknown."""
    try:
        p = torch.cuda.get_device_properties(index)
        bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/local_cpulist") as fh:
            cpus = _parse_cpulist(fh.read())
        return cpus or None
    except Exception:
        return None


def cpus_for_rank(local, nlocal, allowed, gpu_cpus=None):
    """The CPU set of local rank `local` of `nlocal`: the ranks whose GPUs share a NUMA node split that node's allowed CPUs into
    contiguous, disjoint chunks (a rank enqueues ~60 k launches/s: its launch thread must neither migrate nor share a core with
    another rank's); without topology information the allowed CPUs are split evenly.  `gpu_cpus[j]`: CPUs local to GPU j, or None.
    Never returns an empty set (more ranks than CPUs: chunks wrap)."""
    allowed = sorted(allowed)
    if not allowed or nlocal < 1:
        return set(allowed)
    mine = None
    # the NUMA split only with COMPLETE topology: were some ranks to take chunks of their node and the others an even share of everything, the two
    # kinds of chunk could overlap — and "launch threads never share a core" would no longer hold
    if gpu_cpus is not None and (len(gpu_cpus) < nlocal or not all(gpu_cpus[j] and any(c in set(allowed) for c in gpu_cpus[j]) for j in range(nlocal))):
        gpu_cpus = None
    if gpu_cpus is not None and local < len(gpu_cpus) and gpu_cpus[local]:
        mine = tuple(c for c in gpu_cpus[local] if c in set(allowed))
    if mine:
        peers = [j for j in range(nlocal) if j < len(gpu_cpus) and gpu_cpus[j] and
                 tuple(c for c in gpu_cpus[j] if c in set(allowed)) == mine]
        pool, k, m = list(mine), peers.index(local), len(peers)
    else:
        pool, k, m = allowed, local % nlocal, nlocal
    lo, hi = k * len(pool) // m, (k + 1) * len(pool) // m
    return set(pool[lo:hi]) if hi > lo else {pool[k % len(pool)]}


def bind_rank(local, nlocal):
    """Pin this rank to its CPU chunk (see cpus_for_rank); OMNI_BIND_CPUS=0 disables.  Returns the set bound to, or None."""
    if os.environ.get("OMNI_BIND_CPUS", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        allowed = os.sched_getaffinity(0)
        topo = [_gpu_local_cpus(j) for j in range(nlocal)] if torch.cuda.is_available() and torch.cuda.device_count() >= nlocal else None
        cpus = cpus_for_rank(local, nlocal, allowed, topo)
        os.sched_setaffinity(0, cpus)
        # said once per rank: the narrowed set is also what png.effective_cpus(), PngBatches' workers and the decoder pool of this process now see
        if os.environ.get("OMNI_BIND_QUIET", "0") == "0":
            import sys
            print(f"[omnifusion_amd.dist] local rank {local}/{nlocal} bound to {len(cpus)} of {len(allowed)} CPUs "
                  f"({'NUMA-local' if topo and all(topo) else 'even split'}): {_fmt_cpus(cpus)}", file=sys.stderr, flush=True)
        return cpus
    except OSError:
        return None


def init(backend=None):
    """Join the process group the launcher described.  backend None -> "nccl" (= RCCL on ROCm) when a GPU is
    visible, else "gloo".  With nccl, LOCAL_RANK is the device index (one rank per GPU) and the group is bound
    to that device.  Returns (rank, local_rank, world, device)."""
    rank, local, world = env_rank()
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    device = torch.device("cpu")
    if torch.cuda.is_available():
        ndev = torch.cuda.device_count()
        if backend == "nccl" and local >= ndev:
            raise RuntimeError(f"LOCAL_RANK {local} but only {ndev} GPU(s) visible: RCCL needs one GPU per rank")
        local_dev = local % ndev                     # gloo: several ranks may share a GPU (single-GPU test boxes)
        torch.cuda.set_device(local_dev)
        device = torch.device("cuda", local_dev)
    if world > 1 and "LOCAL_WORLD_SIZE" in os.environ:      # under the launcher: one rank per GPU, each on its GPU's NUMA node
        bind_rank(local, int(os.environ["LOCAL_WORLD_SIZE"]))
    # under a launcher (WORLD_SIZE set) the group is created even for ONE rank: the RCCL bring-up, the barrier and the MAX
    # all-reduce of the timing protocol then run on a single-GPU box exactly as they do on eight
    if (world > 1 or "WORLD_SIZE" in os.environ) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world, device


def finalize():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _grouped():
    return dist.is_available() and dist.is_initialized()


def _coll_device(device=None):
    """device a collective's tensors must live on for the active backend"""
    if _grouped() and dist.get_backend() == "nccl":
        return device if device is not None and torch.device(device).type == "cuda" else torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def barrier_sync(device=None):
    """barrier over the ranks, then drain this rank's GPU: both sides of the timed region"""
    if _grouped():
        dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize(device)


def reduce_max(value, device=None):
    """MAX over ranks of a Python float (the job is as slow as its slowest rank)"""
    if not _grouped():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=_coll_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def timed_steps(step_fn, steps, warmup, device=None):
    """bench.py's timing protocol: `warmup` untimed steps, then EXACTLY `steps` calls of step_fn bracketed by
    barrier + device synchronize on both sides; returns the MAX over ranks of the elapsed seconds."""
    for _ in range(warmup):
        step_fn()
    barrier_sync(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    barrier_sync(device)
    return reduce_max(time.perf_counter() - t0, device)


def gather_batch(local_out, n_items):
    """Collect per-rank results (rank r holds the items shard(n_items, r, R), first dimension) into the full
    batch on EVERY rank: one all_gather of equally sized (padded) blocks, then the padding is dropped.  Not on
    the data path — teardown only.  Bitwise: no arithmetic touches the payload."""
    world = _world()
    if world == 1:
        return local_out
    rank = dist.get_rank()
    per = max(shard(n_items, r, world)[1] - shard(n_items, r, world)[0] for r in range(world))
    lo, hi = shard(n_items, rank, world)
    assert local_out.shape[0] == hi - lo, (local_out.shape, lo, hi)
    dev = _coll_device(local_out.device)
    block = torch.zeros((per,) + tuple(local_out.shape[1:]), dtype=local_out.dtype, device=dev)
    block[:hi - lo].copy_(local_out)
    full = torch.empty((world * per,) + tuple(local_out.shape[1:]), dtype=local_out.dtype, device=dev)
    dist.all_gather_into_tensor(full, block)
    parts = []
    for r in range(world):
        a, b = shard(n_items, r, world)
        parts.append(full[r * per:r * per + (b - a)])
    return torch.cat(parts, 0).to(local_out.device)


def broadcast_state_dict(state_dict, src=0):
    """Checkpoint read by rank `src` only, sent to the other ranks (once, at start-up).  `state_dict` may be
    None on the receiving ranks.  Tensors travel through one flat fp32 buffer per dtype class (two large
    messages instead of 375 small ones: xGMI links are per-message latency bound for small payloads)."""
    world = _world()
    if world == 1:
        return state_dict
    rank = dist.get_rank()
    meta = [None]
    if rank == src:
        meta[0] = [(k, tuple(v.shape), str(v.dtype)) for k, v in state_dict.items()]
    dist.broadcast_object_list(meta, src=src)
    dev = _coll_device()
    out = {}
    for cls in (torch.float32, torch.int64):
        items = [(k, s) for k, s, d in meta[0] if (d == "torch.int64") == (cls == torch.int64)]
        n = sum(int(torch.Size(s).numel()) for _, s in items)
        if n == 0:
            continue
        if rank == src:
            flat = torch.cat([state_dict[k].reshape(-1).to(cls) for k, _ in items]).to(dev)
        else:
            flat = torch.empty(n, dtype=cls, device=dev)
        dist.broadcast(flat, src=src)
        flat = flat.cpu()
        o = 0
        for k, s in items:
            m = int(torch.Size(s).numel())
            out[k] = flat[o:o + m].reshape(s).clone()
            o += m
    return {k: out[k] for k, _, _ in meta[0]}


# ---------------------------------------------------------------------------------------------- launcher
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_command(script, script_args, nproc, port=None):
    """The command line that starts `nproc` ranks of `script` on this node — the same form the round driver
    uses (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(nproc)}",
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), script] + list(script_args)


def respawn_under_launcher(script, script_args, nproc, env=None):
    """Run `script` as `nproc` ranks (one per GPU) and return the launcher's exit code.  Used by
    `bench.py --gpus N` when it was started as a plain single process."""
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: required by RCCL on this driver
    e.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(launch_command(script, script_args, nproc), env=e)
