"""Does replaying ag_rollout as a captured HIP graph beat enqueuing it?  ("rollout megaloop as a hipGraph".)
One rollout() call — edges, encoders, three rounds, state update, x steps x streams — eager vs torch.cuda.CUDAGraph replay, at the
reference planner's shape and at C2.     python tools/rollout_graph_probe.py"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adaptigraph_amd import _lib, configs, synth
from adaptigraph_amd.forward_dynamics import rollout
from adaptigraph_amd.graph import threshold_sq
from adaptigraph_amd.model import DynamicsPredictor
dev = torch.device("cuda:0")
w = dict(np.load(os.path.join(ROOT, "tests", "golden", "weights_seed0.npz")))
m = DynamicsPredictor(configs.model_config(), configs.material_config("rope"), configs.dataset_config("rope"), dev)
m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}); m = m.to(dev).eval()
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
for n_obj, B, T in ((100, 500, 15), (200, 500, 15), (200, 64, 15), (1000, 256, 10)):
    g = synth.make_graph_inputs("rope", n_obj, B, seed=0, spacing=0.1)
    thr = threshold_sq(0.5, B, dev, _lib.AG_VARIANT_BATCH)
    rep = torch.full((B,), T, dtype=torch.int32, device=dev)
    args = (m, t(g["state"]), t(g["action"]), t(g["attrs"]), t(g["p_instance"]), t(g["phys"]), t(g["mask"]), t(g["tool_mask"]), thr, rep, T, 10, False, 1)
    for _ in range(3): ref = rollout(*args)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): rollout(*args)
    torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 10 * 1e3
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2): rollout(*args)
    torch.cuda.current_stream().wait_stream(side)
    gr = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(gr):
            out = rollout(*args)
        for _ in range(3): gr.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): gr.replay()
        torch.cuda.synchronize(); graph = (time.perf_counter() - t0) / 10 * 1e3
        same = torch.equal(out, ref)
        print(f"rope {n_obj} particles x {B} samples x {T} steps: eager {eager:.3f} ms, graph replay {graph:.3f} ms ({(graph / eager - 1) * 100:+.1f} %), outputs equal: {same}")
    except Exception as e:      # noqa: BLE001
        print(f"rope {n_obj} x {B} x {T}: eager {eager:.3f} ms, capture failed: {e!r}"[:300])
