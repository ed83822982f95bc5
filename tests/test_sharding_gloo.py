"""CPU, world_size 2 over gloo: the N>1 path of bench.py — contiguous image sharding with no
data-path collective, barrier + max-over-ranks timing, whole-job gather — exercised through the PRODUCT module
omnifusion_amd/dist.py and the launcher command `bench.py --gpus N` re-executes itself under.  The GPU kernels
cannot run here, so the per-rank forward is the CPU oracle; shard equivalence (B images on one rank == the same
images split over 2 ranks, bitwise) is checked on it."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from omnifusion_amd import dist          # noqa: E402


@pytest.mark.parametrize("world,B", [(2, 4), (2, 5), (8, 64)])       # (8, 64): BASELINE cfg 4's literal split, 8 panoramas per rank
def test_two_rank_sharding_matches_single_rank(tmp_path, world, B):
    out = str(tmp_path / "probe.json")
    cmd = dist.launch_command(os.path.join(ROOT, "tests", "_rank_probe.py"), [str(B), out], world)
    assert cmd[1:5] == ["-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}"]
    assert cmd[5:7] == ["--master-addr", "127.0.0.1"]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    info = json.load(open(out))
    assert info["world"] == world and info["shard0"] == list(dist.shard(B, 0, world))
    assert sorted(c for cs in info["cpus"] for c in cs) == sorted(set(c for cs in info["cpus"] for c in cs)) or world > (os.cpu_count() or 1)
    assert all(len(cs) >= 1 for cs in info["cpus"]) and len(info["cpus"]) == world      # every rank bound to a CPU chunk of its own
    # 3 timed steps; the slowest rank sleeps world*10 ms per step: MAX over ranks, not the mean or rank 0's time
    assert info["dt"] >= 3 * 0.01 * world
    from oracle import c_oracle as co
    erp = np.random.default_rng(5).random((B, 1, 32, 64), dtype=np.float32)
    full, _, _, _ = co.equi2pers(erp, 80, 4, 8)
    ref = co.pers2equi(full, 80, 4, 8, (32, 64))
    assert np.array_equal(np.load(out + ".npy"), ref)                 # bitwise shard equivalence, ragged split included


def test_rank_cpu_binding_follows_the_gpu_topology():
    """each rank's launch thread is pinned to a chunk of the CPUs of ITS GPU's NUMA node (sysfs local_cpulist), the ranks
    of one node get disjoint chunks; without topology the allowed CPUs are split evenly; never an empty set."""
    allowed = set(range(64))
    numa = [list(range(0, 32))] * 4 + [list(range(32, 64))] * 4                   # 8 GPUs, 4 per socket
    sets = [dist.cpus_for_rank(r, 8, allowed, numa) for r in range(8)]
    assert all(len(s) == 8 for s in sets) and set().union(*sets) == allowed
    assert all(sets[r] <= set(numa[r]) for r in range(8))
    assert all(sets[a].isdisjoint(sets[b]) for a in range(8) for b in range(a + 1, 8))
    even = [dist.cpus_for_rank(r, 8, allowed, None) for r in range(8)]
    assert even[0] == set(range(8)) and even[7] == set(range(56, 64))
    # PARTIAL topology (one GPU of unknown locality): every rank takes the even split — a node chunk beside an even share would overlap
    part = [dist.cpus_for_rank(r, 2, set(range(4, 12)), [list(range(0, 16)), None]) for r in range(2)]
    assert part[0] == set(range(4, 8)) and part[1] == set(range(8, 12))
    # a cgroup that allows only part of a node: the node's ranks share what is allowed of it
    cg = [dist.cpus_for_rank(r, 2, set(range(4, 12)), [list(range(0, 16)), list(range(0, 16))]) for r in range(2)]
    assert cg[0] == set(range(4, 8)) and cg[1] == set(range(8, 12))
    assert dist._fmt_cpus({0, 1, 2, 3, 8, 10, 11}) == "0-3,8,10-11"
    assert dist.cpus_for_rank(5, 8, {3}, None) == {3} and dist.cpus_for_rank(2, 4, {0, 1}, None) != set()
    assert dist._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]


def test_shard_bounds_cover_batch_exactly():
    for n in (0, 1, 7, 8, 64):
        for w in (1, 2, 4, 8):
            parts = [dist.shard(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in parts) - min(b - a for a, b in parts) <= 1
    with pytest.raises(ValueError):
        dist.shard(8, 2, 2)


def test_single_process_paths_need_no_group():
    """world 1: every helper is the identity / a local operation (bench.py --gpus 1 never creates a process group)"""
    import torch
    calls = []
    dt = dist.timed_steps(lambda: calls.append(1), 4, 2)
    assert len(calls) == 6 and dt >= 0.0
    x = torch.arange(6.0).reshape(3, 2)
    assert dist.gather_batch(x, 3) is x
    assert dist.reduce_max(1.5) == 1.5
    sd = {"w": x}
    assert dist.broadcast_state_dict(sd) is sd


def test_bench_respawns_itself_for_gpus_gt_1(monkeypatch):
    """`python bench.py --gpus N` with WORLD_SIZE unset re-executes under the launcher with N ranks
    (the flag used to be parsed and ignored)."""
    import importlib
    bench = importlib.import_module("bench")
    seen = {}

    def fake(script, script_args, nproc, env=None):
        seen.update(script=script, args=list(script_args), nproc=nproc)
        return 0
    monkeypatch.setattr(dist, "respawn_under_launcher", fake)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    assert seen["nproc"] == 4 and os.path.basename(seen["script"]) == "bench.py"
    assert seen["args"] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
