"""bench.py end to end on the GPU box: the JSON contract, and `--gpus 2` started as ONE process (it must re-execute
itself as 2 ranks; on this single-GPU box the ranks share the GPU through the gloo backend hook — the driver's
8-GPU launch is the same code with backend nccl = RCCL and one GPU per rank)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = dict(os.environ, **(env_extra or {}))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2",
                        "--no-cpu-baseline"] + extra, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # ONE JSON line, from rank 0 only
    return json.loads(lines[0])


def test_bench_json_contract_single_gpu():
    d = _run(["--batch", "2"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert abs(d["value"] - 2 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-6
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert "workload" in d["config"]


def test_bench_gpus_2_spawns_two_ranks():
    d = _run(["--gpus", "2", "--batch", "2"], {"OMNI_BENCH_DIST_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4
    assert abs(d["value"] - 4 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-6


def test_bench_rejects_world_mismatch():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stdout + r.stderr)


def test_bench_under_launcher_one_rank_rccl():
    """The driver's launch form with N = 1: torch.distributed.run starts one rank, the process group is RCCL (backend nccl), and the
    barrier + MAX all-reduce of the timing protocol run on the device — the same calls the 8-GPU launch makes."""
    sys.path.insert(0, ROOT)
    from omnifusion_amd import dist
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "OMNI_BENCH_DIST_BACKEND"):
        env.pop(k, None)
    cmd = dist.launch_command(os.path.join(ROOT, "bench.py"), ["--gpus", "1", "--steps", "3", "--warmup", "2", "--batch", "2", "--no-cpu-baseline"], 1)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0
