"""Stand-in for a rank of bench.py on a box without GPUs: everything except the per-step compute is the PRODUCT
code of omnifusion_amd/dist.py (env parsing, process group, shard, timing protocol, gather, checkpoint
broadcast).  Started by tests/test_sharding_gloo.py through dist.launch_command() — the launcher form the round
driver uses for `bench.py --gpus N`.  The step is the CPU oracle (checker code standing in for the forward)."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from omnifusion_amd import dist          # noqa: E402
from oracle import c_oracle as co        # noqa: E402

B = int(sys.argv[1]); out_path = sys.argv[2]
rank, local, world, dev = dist.init("gloo")
assert dev.type == "cpu" or torch.cuda.is_available()
erp = np.random.default_rng(5).random((B, 1, 32, 64), dtype=np.float32)        # the global batch (same on every rank)
lo, hi = dist.shard(B, rank, world)
state = {}


def step():
    mine, _, _, _ = co.equi2pers(erp[lo:hi], 80, 4, 8)
    state["out"] = co.pers2equi(mine, 80, 4, 8, (32, 64))
    time.sleep(0.01 * (rank + 1))                                               # ranks of unequal speed


dt = dist.timed_steps(step, 3, 1)
full = dist.gather_batch(torch.from_numpy(state["out"]), B)
sd = {"a.weight": torch.arange(6, dtype=torch.float32).reshape(2, 3), "a.n": torch.tensor(7)} if rank == 0 else None
sd = dist.broadcast_state_dict(sd, src=0)
assert sd["a.weight"].tolist() == [[0, 1, 2], [3, 4, 5]] and int(sd["a.n"]) == 7 and list(sd) == ["a.weight", "a.n"]
cpus = [None] * world
torch.distributed.all_gather_object(cpus, sorted(os.sched_getaffinity(0))) if world > 1 else cpus.__setitem__(0, sorted(os.sched_getaffinity(0)))
if rank == 0:
    np.save(out_path + ".npy", full.numpy())
    with open(out_path, "w") as fh:
        json.dump({"world": world, "dt": dt, "shard0": [lo, hi], "cpus": cpus}, fh)
dist.finalize()
