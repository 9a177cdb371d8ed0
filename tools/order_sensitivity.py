"""Per-kernel time of one forward with the particles in their generator's order vs relabelled (gather locality): rope-1k x 256 in polyline
order vs shuffled; granular-2k x 128 (uniform random cloud: the index carries no position) as generated vs sorted by grid cell (0.4 x 0.4, x fastest).
    python tools/order_sensitivity.py"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptigraph_amd import _lib, configs, synth
from adaptigraph_amd.graph import build_edges
from adaptigraph_amd.model import DynamicsPredictor
dev = "cuda:0"
w = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "weights_seed0.npz")))
m = DynamicsPredictor(configs.model_config(), configs.material_config("rope"), configs.dataset_config("rope"), dev)
m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}); m = m.to(dev).eval(); m.set_option("precision", 2)
g = synth.make_graph_inputs("rope", 1000, 256, seed=0, spacing=0.1)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
L = _lib.lib(); h = m.handle(torch.device(dev))
for label, perm in (("ordered", np.arange(1001)), ("shuffled", np.concatenate([np.random.default_rng(0).permutation(1000), [1000]]))):
    st, at, ac, pi = g["state"][:, :, perm], g["attrs"][:, perm], g["action"][:, perm], g["p_instance"][:, perm[:1000]]
    csr = build_edges(t(st[:, -1]), 0.5, t(g["mask"][:, perm]), t(g["tool_mask"][:, perm]), 10, False, "batch", max_tools=1)
    args = (t(st), t(at), csr, None, t(pi)); kw = dict(action=t(ac), rope_physics_param=t(g["phys"]))
    for _ in range(3): m(*args, **kw)
    L.ag_profile_enable(h, 1)
    for _ in range(10): m(*args, **kw)
    ms = (ctypes.c_double * 6)(); cnt = (ctypes.c_int64 * 6)(); e = ctypes.c_int64()
    L.ag_profile_read(h, ms, cnt, ctypes.byref(e)); L.ag_profile_enable(h, 0)
    print(label, {n: round(ms[i] / max(cnt[i], 1), 4) for i, n in enumerate(_lib.KERNEL_CLASSES) if cnt[i]})


# granular: random order (as generated) vs cell order
mg = DynamicsPredictor(configs.model_config(), configs.material_config("granular"), configs.dataset_config("granular"), dev)
mg.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}); mg = mg.to(dev).eval(); mg.set_option("precision", 2)
hg = mg.handle(torch.device(dev))
g = synth.make_graph_inputs("granular", 2000, 128, seed=0)
pos = g["state"][0, -1, :2000]
cell = (np.floor(pos[:, 2] / 0.4) * 1000 + np.floor(pos[:, 0] / 0.4)).astype(np.int64)
order = np.argsort(cell * 10000 + np.argsort(np.argsort(pos[:, 0])), kind="stable")
for label, perm in (("granular as generated", np.arange(2005)), ("granular cell-sorted", np.concatenate([order, np.arange(2000, 2005)]))):
    st, at, ac, pi = g["state"][:, :, perm], g["attrs"][:, perm], g["action"][:, perm], g["p_instance"][:, perm[:2000]]
    csr = build_edges(t(st[:, -1]), 0.4, t(g["mask"][:, perm]), t(g["tool_mask"][:, perm]), 20, False, "batch", max_tools=5)
    args = (t(st), t(at), csr, None, t(pi)); kw = dict(action=t(ac), granular_physics_param=t(g["phys"]))
    for _ in range(3): mg(*args, **kw)
    L.ag_profile_enable(hg, 1)
    for _ in range(10): mg(*args, **kw)
    ms = (ctypes.c_double * 6)(); cnt = (ctypes.c_int64 * 6)(); e = ctypes.c_int64()
    L.ag_profile_read(hg, ms, cnt, ctypes.byref(e)); L.ag_profile_enable(hg, 0)
    print(label, {n: round(ms[i] / max(cnt[i], 1), 4) for i, n in enumerate(_lib.KERNEL_CLASSES) if cnt[i]})
