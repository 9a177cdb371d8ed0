"""TEMPORARY (needs a -DAG_WS_TRACE build, AG_LIB_PATH): s_memtime round timeline of the eight waves of workgroup 3 of edge_encode_ws8_kernel."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["AG_EDGE_WS"] = "2"
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
for _ in range(5): m(*args, **kw)
torch.cuda.synchronize()
L = _lib.lib()
buf = (ctypes.c_uint64 * (8 * 16 * 8 + 8))()
rc = L.ag_ws_trace_read(buf)
allv = np.array(buf, dtype=np.int64); T = allv[:1024].reshape(8, 16, 8)
print('HW_ID per wave: ' + ' '.join('w%d: simd %d cu %d wave %d' % (i, (v >> 4) & 3, (v >> 8) & 15, v & 15) for i, v in enumerate(allv[1024:])))
names = ["RE1 0,1", "RE1 2,3", "RE2 0,1", "RE2 2,3 + L0 4", "Q, L0 0, We 0,1", "Q, L0 1, We 2,3", "RE1 4 | RE2 4", "gather, L0 2,3, We 4"]
print("rc", rc, "rounds 101..115, ticks: [t0 -> each stamp], round = t0(r+1) - t0(r)")
for wv in range(8):
    x = T[wv, 1:15]
    rnd = (T[wv, 2:16, 0] - T[wv, 1:15, 0]).mean()
    d = {k: (x[:, k] - x[:, 0]).mean() for k in range(1, 8) if (x[:, k] > 0).all()}
    print("wave %d (%-18s) round %7.1f  " % (wv, names[wv], rnd) + "  ".join("s%d %7.1f" % (k, v) for k, v in sorted(d.items())))

tb = (ctypes.c_uint64 * (8 * 16 * 8))()
L.ag_ws_tiles_read(tb)
TT = np.array(tb, dtype=np.int64).reshape(8, 16, 8)
for wv in (0, 2, 4, 5):
    x = TT[wv, 1:15]
    print("wave %d phase: per-tile ticks" % wv, " ".join("%7.1f" % (x[:, k + 1] - x[:, k]).mean() for k in range(5)))
