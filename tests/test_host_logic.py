"""Host-side mirror of the reference interface, exercised on CPU (no kernels run here)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import golden_files, load_golden
from adaptigraph_amd import _lib, configs, dist as agdist
from adaptigraph_amd.graph import threshold_sq, csr_from_dense
from adaptigraph_amd.model import DynamicsPredictor, STATE_DICT_KEYS
from adaptigraph_amd.plan_utils import decode_action


@pytest.mark.parametrize("name", golden_files("decode_action"))
def test_decode_action_matches_reference(name):
    g = load_golden(name)
    d, r = decode_action(torch.from_numpy(g["action"]), push_length=float(g["push_length"]))
    assert r.dtype == torch.int32 and np.array_equal(r.numpy(), g["repeat"])
    assert np.abs(d.numpy() - g["decoded"]).max() <= 1e-6


@pytest.mark.parametrize("n_t,grip", [(1, False), (1, True), (5, False), (5, True)])
def test_lean_tool_placement_is_bitwise_the_reference_shaped_one(n_t, grip):
    """dynamics() builds the tool key-points and the per-step motion with _place_tool_lean (one stack each instead of fills and slice
    assignments); dynamics_masked() and the goldens' generator-shaped _place_tool (forward_dynamics.py:42-81) must see the same bits —
    including the sign of zero and non-finite inputs in key-point 0."""
    from adaptigraph_amd import forward_dynamics as fd
    task = {"pusher_points": [[0.0, 0.0]] + [[0.0, 0.013 * i * (-1) ** i] for i in range(1, n_t)], "sim_real_ratio": 10.0, "gripper_enable": grip}
    g = torch.Generator().manual_seed(5)
    action = torch.rand((9, 1, 4), generator=g) * 6 - 3
    action[0, 0, 0], action[1, 0, 1] = -0.0, float("inf")
    decoded, _ = decode_action(action, push_length=0.1)
    y = torch.randn(9, generator=g)
    eef, dlt, up = fd._place_tool(task, decoded[:, 0], action[:, 0, 2], y, "cpu")
    eef2, dlt2, up2 = fd._place_tool_lean(task, decoded[:, 0], action[:, 0, 2], y, "cpu")
    same = lambda a, b: torch.equal(a.view(torch.int32), b.expand_as(a).contiguous().view(torch.int32))      # noqa: E731  (bit patterns: -0.0, NaN)
    assert up == up2 and same(eef, eef2) and same(dlt, dlt2)


def test_threshold_rounding_per_variant():
    dev = torch.device("cpu")
    # 0.4: fp32(0.4)^2 and fp32(0.4^2 in double) differ by one ulp
    single = threshold_sq(0.4, 2, dev, _lib.AG_VARIANT_SINGLE)
    batch = threshold_sq(0.4, 2, dev, _lib.AG_VARIANT_BATCH)
    assert single[0].item() == float(np.float32(0.4 * 0.4))
    assert batch[0].item() == float(np.float32(0.4) * np.float32(0.4))
    assert single[0].item() != batch[0].item()
    per = threshold_sq(torch.tensor([0.3, 0.5]), 2, dev, _lib.AG_VARIANT_BATCH)
    assert torch.equal(per, torch.tensor([0.3, 0.5]) ** 2)


def test_state_dict_layout_is_the_reference_checkpoint_layout(weights):
    m = DynamicsPredictor(configs.model_config(), configs.material_config("rope"), configs.dataset_config("rope"), "cpu")
    sd = m.state_dict()
    assert list(sd.keys()) == STATE_DICT_KEYS == list(weights.keys())
    assert sum(v.numel() for v in sd.values()) == 252903
    for k in STATE_DICT_KEYS:
        assert tuple(sd[k].shape) == weights[k].shape
    m.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})     # strict
    assert torch.equal(m.relation_propagator.linear.weight, torch.from_numpy(weights["relation_propagator.linear.weight"]))


def test_unsupported_configs_raise_like_the_reference():
    mc = configs.model_config()
    mc["offset_dim"] = 3
    with pytest.raises(NotImplementedError):
        DynamicsPredictor(mc, configs.material_config("rope"), configs.dataset_config("rope"), "cpu")
    two = configs.material_config("rope")
    two["material_index"]["cloth"] = 1
    with pytest.raises(AssertionError):
        DynamicsPredictor(configs.model_config(), two, configs.dataset_config("rope"), "cpu")


def test_forward_refuses_cpu_tensors(weights):
    m = DynamicsPredictor(configs.model_config(), configs.material_config("rope"), configs.dataset_config("rope"), "cpu")
    g = load_golden("fwd_rope64")
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.from_numpy(g["state"]), torch.from_numpy(g["attrs"]), None, None, torch.from_numpy(g["p_instance"]),
          action=torch.from_numpy(g["action"]), rope_physics_param=torch.from_numpy(g["phys"]))


def test_csr_from_dense_drops_padding_and_sorts_by_receiver():
    N = 5
    recv = [3, 0, 3, 1]
    send = [1, 2, 4, 0]
    Rr = torch.zeros(1, 7, N)
    Rs = torch.zeros(1, 7, N)
    for e, (r, s) in enumerate(zip(recv, send)):
        Rr[0, e, r] = 1
        Rs[0, e, s] = 1
    csr = csr_from_dense(Rr, Rs)                 # 3 padded all-zero rows vanish (utils.py:127-137)
    assert csr.row_ptr.tolist() == [0, 1, 2, 2, 4, 4]
    assert csr.edge_recv[:4].tolist() == [0, 1, 3, 3]
    assert csr.edge_send[:4].tolist() == [2, 0, 1, 4]          # stable within a receiver


def test_shard_bounds_cover_the_batch():
    for total, world in ((1024, 8), (10, 4), (3, 8), (256, 1)):
        seen = []
        for r in range(world):
            lo, hi, per = agdist.shard_bounds(total, r, world)
            assert hi - lo <= per
            seen += list(range(lo, hi))
        assert seen == list(range(total))


def _gloo_worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        state = torch.arange(12, dtype=torch.float32).reshape(4, 3)
        action = torch.arange(total * 2 * 4, dtype=torch.float32).reshape(total, 2, 4)
        calls = []

        def fake_dynamics(st, act, tag):      # stands in for the engine: result depends only on the sample's action
            calls.append(act.shape[0])
            seq = act.sum(-1)[:, :, None, None] + st[None, None]
            return {"state_seqs": seq, "action_seqs": act * 2}

        out = agdist.dynamics_sharded(fake_dynamics, state, action, "x")
        local = list(calls)                   # the shard this rank rolled out (nothing at all when there are more ranks than samples)
        full = fake_dynamics(state, action, "x")
        lo, hi, per = agdist.shard_bounds(total, rank, world)
        ok = all(torch.equal(out[k], full[k]) for k in full) and local == ([hi - lo] if hi > lo else []) and hi - lo <= per
        # a second call reuses the cached collective buffers; the first result must not be overwritten (copy semantics)
        out2 = agdist.dynamics_sharded(fake_dynamics, state, action + 1.0, "x")
        full2 = fake_dynamics(state, action + 1.0, "x")
        ok = ok and all(torch.equal(out2[k], full2[k]) for k in full2) and all(torch.equal(out[k], full[k]) for k in full)
        # copy=False (bench / MPPI path): views of the cached receive buffers — right contents, one buffer per key, overwritten by the next gather
        out3 = agdist.dynamics_sharded(fake_dynamics, state, action + 2.0, "x", copy=False)
        full3 = fake_dynamics(state, action + 2.0, "x")
        ok = ok and all(torch.equal(out3[k], full3[k]) for k in full3) and out3["state_seqs"].data_ptr() != out3["action_seqs"].data_ptr()
        out4 = agdist.dynamics_sharded(fake_dynamics, state, action + 3.0, "x", copy=False)
        ok = ok and out4["state_seqs"].data_ptr() == out3["state_seqs"].data_ptr() and all(torch.equal(out2[k], full2[k]) for k in full2)
        # replicated-input contract: per-rank draws differ until replicate() broadcasts rank 0's (MPPI samples)
        torch.manual_seed(100 + rank)
        draw = torch.rand(total, 2, 4)
        try:
            agdist.assert_replicated(draw, "draw")
            ok = False
        except RuntimeError:
            pass
        draw = agdist.replicate(draw)
        agdist.assert_replicated(draw, "draw")
        torch.manual_seed(100)
        ok = ok and torch.equal(draw, torch.rand(total, 2, 4))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def _run_gloo(world, total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 7 + world * 131 + total) % 3000
    procs = [ctx.Process(target=_gloo_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, True) for r in range(world)]


@pytest.mark.parametrize("total", [8, 5, 1])
def test_dynamics_sharded_gloo_world2(total):
    _run_gloo(2, total)


@pytest.mark.parametrize("world,total", [(4, 10), (8, 1024), (8, 500), (8, 3)])
def test_dynamics_sharded_gloo_world4_and_8(world, total):
    """The SCALE run's shapes before the first real RCCL contact: BASELINE configs[4]'s 1024 samples over 8 ranks (even), 500 over 8
    (uneven: seven shards of 63 and one of 59), fewer samples than ranks (empty shards), and a 4-rank uneven split."""
    _run_gloo(world, total)


def _torchrun(script, n, extra, timeout=600):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 29100 + (os.getpid() * 13 + n * 7 + len(extra)) % 700
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, script), "--gpus", str(n), "--dry-run"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=dict(os.environ, OMP_NUM_THREADS="2"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    return json.loads(lines[0])


def test_bench_dry_run_world8_strong_scaling_contract():
    """The driver's launch line for the SCALE run, N = 8, BASELINE configs[3] in its strong-scaling form (--global-batch 512): process group,
    sharding, all-gather, max-over-ranks timing and the JSON line with everything but the engine (bench.py --dry-run: gloo, CPU tensors, a
    stand-in rollout whose gathered result is checked against the unsharded one inside the run).  No GPU, no number: the key contract only."""
    d = _torchrun("bench.py", 8, ["--steps", "2", "--warmup", "0", "--material", "cloth", "--global-batch", "512", "--rollout-steps", "20"])
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["steps"] == 2 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["unit"] == "graph-steps/s" and d["data"].startswith("dry-run")
    cfg = d["config"]
    assert cfg["global_batch"] == 512 and cfg["rollout_steps"] == 20 and cfg["parallelism"] == "batch-shard x8 + all-gather"
    assert "batch 64/GPU" in cfg["workload"] and "global batch 512 over 8 GPUs" in cfg["workload"]
    assert all(isinstance(v, (int, float, str)) and (not isinstance(v, str) or len(v) <= 100) for v in cfg.values()), "flat, short config values"
    assert set(d["ranks"]) >= {"ms_per_step", "rollout_ms", "all_gather_ms"} and d["ranks"]["ms_per_step"]["max"] >= d["ranks"]["ms_per_step"]["min"]
    assert abs(d["value"] - 512 * 20 * 2 / (d["ms_per_step"] * 2 / 1e3)) <= 1e-6 * d["value"]


def test_bench_dry_run_world8_weak_scaling_and_mpc_contract():
    """Weak-scaling form (the driver's default: per-GPU batch fixed) and bench_mpc.py (BASELINE configs[4]: 1024 samples over 8 ranks)."""
    d = _torchrun("bench.py", 8, ["--steps", "1", "--warmup", "0", "--batch", "4"])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["config"]["global_batch"] == 32 and "batch 4/GPU" in d["config"]["workload"]
    m = _torchrun("bench_mpc.py", 8, ["--steps", "1", "--warmup", "1", "--samples", "1024", "--particles", "50"])
    assert m["n_gpus"] == 8 and m["scaling"] == "strong" and m["higher_is_better"] is False and m["unit"] == "ms"
    assert m["config"]["samples"] == 1024 and m["config"]["parallelism"] == "samples/8" and m["value"] > 0
