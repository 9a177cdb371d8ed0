"""Round timeline of the weight-stationary edge encoder (debug build -DAG_TRACE=1): per wave, s_memtime deltas of
   1 round start -> 2 first phase done -> 3 second phase done (at the barrier) -> 4 barrier passed, rounds 100..107 of workgroup 3.
   AG_LIB_PATH=ab/libtrace.so python tools/trace_ws.py"""
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
csr = build_edges(t(g["state"][:, -1]), 0.5, t(g["mask"]), t(g["tool_mask"]), 10, False, "batch", max_tools=1)
args = (t(g["state"]), t(g["attrs"]), csr, None, t(g["p_instance"]))
kw = dict(action=t(g["action"]), rope_physics_param=t(g["phys"]))
for _ in range(3): m(*args, **kw)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 2048)()
assert _lib.lib().ag_debug_trace_read(buf) == 0
a = np.array(buf, dtype=np.uint64).reshape(4, 256, 2).astype(np.int64)
t0 = min(int(a[s][0, 1]) for s in range(4) if a[s][0, 0] != 0)
for s in range(4):
    st = a[s]; n = int((st[:, 0] != 0).sum()); st = st[:n]
    if n == 0: continue
    line, prev = [], None
    for tag, tm in st:
        if tag == 1: line.append(f"| @{tm - t0}")
        else: line.append(f"{tag}:+{tm - prev}")
        prev = tm
    print(f"wave {s}: " + " ".join(line))
