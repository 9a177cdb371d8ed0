"""Race hunt: the same forward / rollout repeated many times at the full C2 shape must be bitwise identical every time
(all reductions have fixed orders; the weight ring, the counted waits and the tile queue must never change a result)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptigraph_amd import configs, synth
from adaptigraph_amd.forward_dynamics import dynamics
from adaptigraph_amd.graph import build_edges
from adaptigraph_amd.model import DynamicsPredictor
dev = "cuda:0"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
w = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "weights_seed0.npz")))
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
bad = 0
for prec in (2, 1, 0):
    m = DynamicsPredictor(configs.model_config(), configs.material_config("rope"), configs.dataset_config("rope"), dev)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}); m = m.to(dev).eval(); m.set_option("precision", prec)
    g = synth.make_graph_inputs("rope", 1000, 256, seed=0, spacing=0.1)
    csr = build_edges(t(g["state"][:, -1]), 0.5, t(g["mask"]), t(g["tool_mask"]), 10, False, "batch", max_tools=1)
    args = (t(g["state"]), t(g["attrs"]), csr, None, t(g["p_instance"]))
    kw = dict(action=t(g["action"]), rope_physics_param=t(g["phys"]))
    ref = m(*args, **kw)[1].clone()
    n = reps if prec == 2 else reps // 4
    for i in range(n):
        if not torch.equal(m(*args, **kw)[1], ref):
            bad += 1
            print(f"prec {prec}: forward {i} differs")
    print(f"prec {prec}: {n} forwards compared")
    ppm = configs.ppm_optimizer_stub("rope"); ppm.physics_param = {"rope": torch.tensor([0.5], device=dev)}
    state, act = synth.make_mpc_inputs("rope", 1000, 256, seed=0, len_lo=6, len_hi=6.9, spacing=0.1)
    r0 = dynamics(t(state), t(act), m, dev, ppm)["state_seqs"].clone()
    n = max(1, reps // 10) if prec == 2 else max(1, reps // 40)
    for i in range(n):
        if not torch.equal(dynamics(t(state), t(act), m, dev, ppm)["state_seqs"], r0):
            bad += 1
            print(f"prec {prec}: rollout {i} differs")
    print(f"prec {prec}: {n} rollouts (2 streams, 6 steps) compared")
print("MISMATCHES:", bad)
sys.exit(1 if bad else 0)
