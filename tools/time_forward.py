"""Per-kernel timing of ONE forward on a fixed graph (rope-1k, batch 256) — for kernel A/B builds.
   AG_LIB_PATH=/path/to/variant.so python tools/time_forward.py [precision] [reps]"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptigraph_amd import _lib, configs, synth
from adaptigraph_amd.graph import build_edges
from adaptigraph_amd.model import DynamicsPredictor
prec = int(sys.argv[1]) if len(sys.argv) > 1 else 2
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = "cuda:0"
w = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "weights_seed0.npz")))
m = DynamicsPredictor(configs.model_config(), configs.material_config("rope"), configs.dataset_config("rope"), dev)
m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}); m = m.to(dev).eval(); m.set_option("precision", prec)
g = synth.make_graph_inputs("rope", 1000, 256, seed=0, spacing=0.1)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
csr = build_edges(t(g["state"][:, -1]), 0.5, t(g["mask"]), t(g["tool_mask"]), 10, False, "batch", max_tools=1)
args = (t(g["state"]), t(g["attrs"]), csr, None, t(g["p_instance"]))
kw = dict(action=t(g["action"]), rope_physics_param=t(g["phys"]))
for _ in range(3): m(*args, **kw)
L = _lib.lib(); h = m.handle(torch.device(dev))
L.ag_profile_enable(h, 1)
for _ in range(reps): m(*args, **kw)
ms = (ctypes.c_double * 6)(); cnt = (ctypes.c_int64 * 6)(); e = ctypes.c_int64()
L.ag_profile_read(h, ms, cnt, ctypes.byref(e))
print(os.environ.get("AG_LIB_PATH", "default"), {n: round(ms[i] / max(cnt[i], 1), 4) for i, n in enumerate(_lib.KERNEL_CLASSES) if cnt[i]})
