"""Latency of ONE small forward with the node encoder per node / de-duplicated / auto (ag_set_option("node_dedup", 0 / 2 / 1)):
why the default only de-duplicates calls of >= 32 768 node-rows x steps.   python tools/small_forward_latency.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from adaptigraph_amd import configs, synth
from adaptigraph_amd.graph import build_edges
from adaptigraph_amd.model import DynamicsPredictor
dev="cuda:0"
w = dict(np.load(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests", "golden", "weights_seed0.npz")))
m = DynamicsPredictor(configs.model_config(), configs.material_config("rope"), configs.dataset_config("rope"), dev)
m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}); m = m.to(dev).eval()
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
for n_obj, B in ((100, 1), (100, 6), (300, 1), (1000, 1), (1000, 8), (200, 64)):
    g = synth.make_graph_inputs("rope", n_obj, B, seed=0, spacing=0.1)
    csr = build_edges(t(g["state"][:, -1]), 0.5, t(g["mask"]), t(g["tool_mask"]), 10, False, "batch", max_tools=1)
    args = (t(g["state"]), t(g["attrs"]), csr, None, t(g["p_instance"])); kw = dict(action=t(g["action"]), rope_physics_param=t(g["phys"]))
    res = {}
    for dd in (0, 2, 1):
        m.set_option("node_dedup", dd)
        for _ in range(5): m(*args, **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): m(*args, **kw)
        torch.cuda.synchronize(); res[dd] = (time.perf_counter() - t0) / 50 * 1e3
    print(f"rope n_obj={n_obj} B={B}: forward {res[0]:.3f} ms per-node encoder, {res[2]:.3f} ms de-duplicated always, {res[1]:.3f} ms default (auto)")
