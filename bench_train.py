#!/usr/bin/env python3
"""Side benchmark for the training path: wall-clock of one optimisation step of the reference's
training objective (n_future = 3 unrolled forward passes + backward + Adam) at the reference's training shape
(config/dynamics/rope.yaml: batch 128, <= 100 key-points + 1 tool per graph, <= 1000 relations), on one MI355X.

    python bench_train.py [--batch 128] [--steps 20] [--warmup 5] [--dense-baseline]

`value` times `TrainableDynamicsPredictor` (CSR gather / segment-reduce HIP kernels + library GEMMs).  With
--dense-baseline the same step is also timed in the reference's formulation — one-hot Rr/Rs matrices and `bmm`, restated
here in plain torch on the same GPU (`baseline_dense_bmm_ms`) — with the loss of both checked to agree.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from adaptigraph_amd import configs, graph as aggraph                          # noqa: E402
from adaptigraph_amd.train_model import TrainableDynamicsPredictor, unrolled_loss   # noqa: E402
from adaptigraph_amd.train_ops import EdgeViews                                 # noqa: E402


def synthetic_batch(B, n_obj_max, dev, seed=0):
    """Rope-like key-point clouds with 60-100 valid points per sample, padded to n_obj_max + 1 tool slot."""
    rng = np.random.default_rng(seed)
    N = n_obj_max + 1
    state = np.zeros((B, 4, N, 3), np.float32)
    mask = np.zeros((B, N), bool)
    for b in range(B):
        n = int(rng.integers(60, n_obj_max + 1))
        i = np.arange(n)
        cur = np.stack([i * 0.2, np.zeros(n), 2.0 * np.sin(2 * np.pi * i / n)], 1) + rng.normal(0, 0.02, (n, 3))
        for h in range(4):
            state[b, h, :n] = cur + rng.normal(0, 0.01, (n, 3)) * (3 - h)
            state[b, h, -1] = [cur[n // 2, 0], 0.0, cur[n // 2, 2] + 0.3 - 0.05 * (3 - h)]
        mask[b, :n] = True
        mask[b, -1] = True
    tool = np.zeros((B, N), bool)
    tool[:, -1] = True
    attrs = np.zeros((B, N, 2), np.float32)
    attrs[..., 0] = mask & ~tool
    attrs[..., 1] = tool
    action = np.zeros((B, N, 3), np.float32)
    action[:, -1] = [0.0, 0.0, 0.05]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    data = {"state": t(state), "attrs": t(attrs), "action": t(action), "p_instance": t(attrs[:, :n_obj_max, :1].copy()),
            "rope_physics_param": torch.full((B, 1), 0.5, device=dev),
            "state_future": t(state[:, -1:, :n_obj_max].repeat(3, 1) + rng.normal(0, 0.02, (B, 3, n_obj_max, 3)).astype(np.float32)),
            "eef_future": t(np.repeat(state[:, -1:], 2, 1) * tool[:, None, :, None]), "action_future": t(np.repeat(action[:, None], 2, 1))}
    csr = aggraph.build_edges(t(state[:, -1]), 0.5, t(mask), t(tool), 10, False, "single", max_tools=1)
    return data, csr


def dense_forward(model, state, attrs, Rr, Rs, p_instance, action, phys):
    """The reference formulation (model.py:129-313): one-hot relation matrices and bmm (oracle/torch_dense.py)."""
    from oracle.torch_dense import dense_forward as f
    return f(model, state, attrs, Rr, Rs, p_instance, action, phys)[0]


def dense_unrolled_loss(model, data, Rr, Rs, n_future=3):
    state, action, loss = data["state"], data["action"], 0
    for fi in range(n_future):
        pred = dense_forward(model, state, data["attrs"], Rr, Rs, data["p_instance"], action, data["rope_physics_param"])
        loss = loss + F.mse_loss(pred, data["state_future"][:, fi])
        if fi < n_future - 1:
            nxt = data["eef_future"][:, fi].clone()
            nxt[:, :pred.shape[1]] = pred
            state, action = torch.cat([state[:, 1:], nxt[:, None]], 1), data["action_future"][:, fi]
    return loss


def timed(step, steps, warmup):
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--max-nobj", type=int, default=100)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dense-baseline", action="store_true")
    ap.add_argument("--library-gemm", action="store_true", help="dense stacks through torch F.linear (the r01 path) instead of the fused MFMA chains")
    ap.add_argument("--graph", action="store_true", help="capture the whole optimisation step in a HIP graph (torch.cuda.graphs) and replay it")
    ap.add_argument("--autograd-grads", action="store_true", help="parameter gradients through autograd's accumulation instead of straight into .grad")
    a = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("bench_train.py needs an MI355X: the graph kernels have no CPU path")
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = TrainableDynamicsPredictor(configs.model_config(), configs.material_config("rope"), configs.dataset_config("rope"), dev).to(dev).train()
    model.fused_dense = not a.library_gemm
    from adaptigraph_amd import train_ops
    direct = model.fused_dense and not a.autograd_grads     # parameter gradients straight into .grad, scoped to the training steps below
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, capturable=a.graph)
    data, csr = synthetic_batch(a.batch, a.max_nobj, dev)
    data.update(Rr=csr, Rs=None, edge_views=EdgeViews(csr))

    def step():
        opt.zero_grad(set_to_none=False)     # keep the .grad buffers: the gradient kernel accumulates into them
        with train_ops.direct_grads(direct):
            loss = unrolled_loss(model, data, 3)
            loss.backward()
        opt.step()
        return loss

    init = {k: v.clone() for k, v in model.state_dict().items()}
    if a.graph:
        # The step has no host synchronisation (edge views are per batch, resolved above), so it captures as ONE HIP graph:
        # ~300 launches replayed without their host-side launch cost.  Warm up on a side stream first (allocator, packed
        # weight streams, Adam state), as torch.cuda.graphs requires.
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        opt.zero_grad(set_to_none=False)
        with torch.cuda.graph(g):
            opt.zero_grad(set_to_none=False)              # inside the graph: every replay starts from zero gradients
            with train_ops.direct_grads(direct):
                static_loss = unrolled_loss(model, data, 3)
                static_loss.backward()
            opt.step()
        eager_step = step

        def step():                   # noqa: F811
            g.replay()
            return static_loss
    ms, _ = timed(step, a.steps, a.warmup)
    line = {"metric": "training step wall-clock (3-step unroll forward + backward + Adam)", "value": round(ms, 3), "unit": "ms", "n_gpus": 1,
            "steps": a.steps, "warmup": a.warmup, "higher_is_better": False,
            "dtype": "f32" if (a.library_gemm or not train_ops.CHAIN_PRECISION) else "bf16x3", "data": "synthetic",
            "config": {"workload": f"rope key-point graphs, batch {a.batch}, <= {a.max_nobj}+1 nodes, {int(csr.n_rel().sum())} edges in the batch"},
            "graphs_per_s": round(a.batch / ms * 1e3, 1),
            "dense_stacks": "library GEMMs (F.linear)" if a.library_gemm else ("fused MFMA chain kernels (forward + backward), " + ("split-bf16 operands (3 bf16 products, f32 accumulate)" if train_ops.CHAIN_PRECISION else "exact f32 MFMA")),
            "hip_graph": bool(a.graph), "direct_grads": bool(direct)}
    train_ops.invalidate_packs()      # graph replays updated the parameters behind the version counters the pack cache keys on
    if a.dense_baseline:
        Rr, Rs = csr.to_dense(torch.float32)
        model.load_state_dict(init)
        l_csr = unrolled_loss(model, data, 3).item()
        l_dense = dense_unrolled_loss(model, data, Rr, Rs).item()
        assert abs(l_csr - l_dense) <= 1e-5 * max(1.0, abs(l_dense)), (l_csr, l_dense)
        opt2 = torch.optim.Adam(model.parameters(), lr=1e-3)

        def dense_step():
            opt2.zero_grad()
            loss = dense_unrolled_loss(model, data, Rr, Rs)
            loss.backward()
            opt2.step()
            return loss

        ms_d, _ = timed(dense_step, a.steps, a.warmup)
        line.update(baseline_dense_bmm_ms=round(ms_d, 3), speedup_vs_dense_bmm=round(ms_d / ms, 2), loss_check=[l_csr, l_dense])
    print(json.dumps(line))


if __name__ == "__main__":
    main()
