"""Multi-GPU readiness that one leased GPU can check (VERDICT r01 #7a): a 1-rank "nccl" (= RCCL) process group on the device —
RCCL initialises, collectives run on device memory, HSA_ENABLE_IPC_MODE_LEGACY handling is in place — driving the same
code paths bench.py --gpus N uses (dist.gather_sharded / replicate / assert_replicated, the timing hooks).  The N > 1 data
path itself is covered on CPU with world-size-2 gloo (tests/test_host_logic.py); no scaling number is claimed here."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
sys.path.insert(0, os.environ["AG_ROOT"])
import numpy as np, torch, torch.distributed as dist
from adaptigraph_amd import configs, synth, dist as agdist
from adaptigraph_amd.forward_dynamics import dynamics
from adaptigraph_amd.model import DynamicsPredictor
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
dist.init_process_group("nccl", device_id=dev)                      # RCCL
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
w = dict(np.load(os.path.join(os.environ["AG_ROOT"], "tests", "golden", "weights_seed0.npz")))
m = DynamicsPredictor(configs.model_config(), configs.material_config("rope"), configs.dataset_config("rope"), dev)
m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}); m = m.to(dev).eval()
ppm = configs.ppm_optimizer_stub("rope"); ppm.physics_param = {"rope": torch.tensor([0.5], device=dev)}
state, act = synth.make_mpc_inputs("rope", 200, 6, seed=2, len_lo=2, len_hi=3.9, spacing=0.1)
state, act = torch.from_numpy(state).to(dev), torch.from_numpy(act).to(dev)
act = agdist.replicate(act)                                          # broadcast on device memory
agdist.assert_replicated(act, "act")                                 # all_reduce MIN/MAX on device memory
local = dynamics(state, act, m, dev, ppm)
lo, hi, per = agdist.shard_bounds(act.shape[0], 0, 1)
timing = {}
out = {k: agdist.gather_sharded(v, act.shape[0], per) for k, v in local.items()}     # all_gather_into_tensor through RCCL
again = {k: agdist.gather_sharded(v, act.shape[0], per, copy=False) for k, v in local.items()}
assert all(torch.equal(out[k], local[k]) and torch.equal(again[k], local[k]) for k in local)
full = agdist.dynamics_sharded(dynamics, state, act, m, dev, ppm, timing=timing)     # world 1: plain call, no events
assert torch.equal(full["state_seqs"], local["state_seqs"]) and timing == {}
dist.barrier(); torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL_SINGLE_RANK_OK")
"""


def test_single_rank_rccl_group_runs_the_sharded_path():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29400 + os.getpid() % 500),
               AG_ROOT=ROOT)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", WORKER], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_SINGLE_RANK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_bench_entrypoint_under_torchrun_single_rank():
    """bench.py launched exactly as the driver launches it for N > 1 (torch.distributed.run), with one rank: rendezvous on
    127.0.0.1, RANK/LOCAL_RANK/WORLD_SIZE from the environment, one JSON line from rank 0."""
    import json
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(29900 + os.getpid() % 90), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--batch", "32", "--no-cpu-baseline", "--no-extra"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-4000:]
    line = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["roofline"]["frac"] > 0 and line["roofline"]["mfma_issue_util"] > line["roofline"]["frac"]


def test_bench_two_ranks_one_gpu_over_gloo():
    """The N > 1 bench path end to end with the real engine: two ranks launched exactly like the driver launches them, both on
    the one leased GPU, the collective over gloo instead of RCCL (two RCCL ranks cannot share a device).  Checks the contract of
    the JSON line for N = 2 (global batch, weak scaling, per-rank timings) — not a scaling number."""
    import json
    env = dict(os.environ, AG_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29700 + os.getpid() % 90), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "24", "--no-cpu-baseline", "--no-profile"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 48 and line["scaling"] == "weak" and line["value"] > 0
    rk = line["ranks"]
    assert rk["ms_per_step"]["max"] >= rk["ms_per_step"]["min"] > 0 and rk["all_gather_ms"]["max"] > 0 and rk["rollout_ms"]["min"] > 0
    assert "extra" not in line and "cpu_baseline" not in line


def test_bench_strong_scaling_global_batch_two_ranks_over_gloo():
    """`bench.py --global-batch` (BASELINE configs[3]'s mode: a fixed global batch split over the ranks) as the driver would launch
    it at N = 2, cloth workload, roofline objects present on the N > 1 line, scaling reported as strong."""
    import json
    env = dict(os.environ, AG_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29600 + os.getpid() % 90), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--material", "cloth", "--global-batch", "6", "--rollout-steps", "3", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["global_batch"] == 6 and line["value"] > 0
    assert "3/GPU" in line["config"]["workload"] and line["roofline"]["frac"] > 0 and line["roofline_hbm"]["frac"] > 0


def test_mpc_bench_two_ranks_one_gpu_over_gloo():
    """bench_mpc.py (BASELINE configs[4]) with the samples sharded over two ranks: rank 0's sampled actions are broadcast, each
    rank rolls out its half, the states are all-gathered and every rank runs the cost + MPPI update on the full set.  The
    per-iteration time must come out of rank 0 as one JSON line; the planner's result must not depend on the sharding (the
    2-rank reward vector equals the 1-rank one is covered by the bitwise chunk-equality test of the rollout)."""
    import json
    env = dict(os.environ, AG_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29800 + os.getpid() % 90), os.path.join(ROOT, "bench_mpc.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--particles", "200", "--samples", "64", "--push-steps", "4"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0 and line["config"]["parallelism"] == "samples/2"


def test_mpc_bench_eight_ranks_one_gpu_over_gloo():
    """The argument / JSON contract of `bench_mpc.py --gpus 8` (BASELINE configs[4] on a full node), with the eight ranks on the one leased
    GPU and the collective over gloo: 60 samples do not divide by 8 (ranks get 8, 8, 8, 8, 7, 7, 7, 7), every rank must reach the
    all-gather, rank 0 alone prints the line."""
    import json
    env = dict(os.environ, AG_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(29500 + os.getpid() % 90), os.path.join(ROOT, "bench_mpc.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
           "--particles", "100", "--samples", "60", "--push-steps", "3"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["scaling"] == "strong" and line["value"] > 0 and line["higher_is_better"] is False
    assert line["config"]["parallelism"] == "samples/8" and line["config"]["samples"] == 60 and line["steps"] == 2 and line["warmup"] == 1
