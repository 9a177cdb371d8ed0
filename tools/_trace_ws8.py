"""TEMPORARY (needs a -DAG_WS_TRACE build, AG_LIB_PATH): s_memtime round timeline of the eight waves of workgroup 3 of edge_encode_ws8_kernel."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["AG_EDGE_WS"] = "1"
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
names = ["RE1 0,1 + L0 0", "RE1 2,3 + L0 1", "epi, RE2 0,1", "epi, RE2 2,3", "Q, L0 2, We 0,1", "Q, L0 3, We 2,3", "RE1 4, RE2 4", "We 4, Q, gather, L0 4"]
print("rc", rc, "rounds 101..115, ticks: [t0 -> each stamp], round = t0(r+1) - t0(r)")
for wv in range(8):
    x = T[wv, 1:15]
    rnd = (T[wv, 2:16, 0] - T[wv, 1:15, 0]).mean()
    d = {k: (x[:, k] - x[:, 0]).mean() for k in range(1, 8) if (x[:, k] > 0).all()}
    print("wave %d (%-18s) round %7.1f  " % (wv, names[wv], rnd) + "  ".join("s%d %7.1f" % (k, v) for k, v in sorted(d.items())))

