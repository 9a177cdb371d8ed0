"""Streamed segment reduce (aggregate_stream_kernel) vs aggregate_half_kernel: bitwise equality of a forward and per-launch time.
   AG_LIB_PATH=/path/to/variant.so python tools/agg_check.py [material=rope] [n_obj=1000] [batch=256] [reps=20]"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptigraph_amd import _lib, configs, synth
from adaptigraph_amd.graph import build_edges
from adaptigraph_amd.model import DynamicsPredictor
material = sys.argv[1] if len(sys.argv) > 1 else "rope"
n_obj = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 256
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
dev = "cuda:0"
w = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "weights_seed0.npz")))
m = DynamicsPredictor(configs.model_config(), configs.material_config(material), configs.dataset_config(material), dev)
m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}); m = m.to(dev).eval(); m.set_option("precision", 2)
kw = dict(spacing=0.1) if material == "rope" else {}
g = synth.make_graph_inputs(material, n_obj, batch, seed=0, **kw)
mm = synth.MATERIALS[material]
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
csr = build_edges(t(g["state"][:, -1]), mm["radius"], t(g["mask"]), t(g["tool_mask"]), mm["topk"], mm["connect_tools_all"], "batch", max_tools=g["n_tools"])
args = (t(g["state"]), t(g["attrs"]), csr, None, t(g["p_instance"]))
kwp = {"action": t(g["action"]), material + "_physics_param": t(g["phys"])}
L = _lib.lib(); h = m.handle(torch.device(dev))
out = {}
for mode in (0, 1):
    m.set_option("aggregate_stream", mode)
    for _ in range(3): pos, mot = m(*args, **kwp)
    torch.cuda.synchronize()
    out[mode] = (pos.clone(), mot.clone())
    L.ag_profile_enable(h, 1)
    for _ in range(reps): m(*args, **kwp)
    ms = (ctypes.c_double * 6)(); cnt = (ctypes.c_int64 * 6)(); e = ctypes.c_int64()
    L.ag_profile_read(h, ms, cnt, ctypes.byref(e))
    L.ag_profile_enable(h, 0)
    if mode and hasattr(L, "ag_debug_agg_trace"):
        tr = (ctypes.c_ulonglong * 16)(); L.ag_debug_agg_trace(tr, 1)
        n = max(cnt[3], 1) + 9      # launches incl. warm-up (3 forwards x 3 rounds)
        print("trace per launch: loader cyc %.0f (in vmcnt waits %.0f, %d waits, %d by depth, %d idle sleeps) | consumer cyc %.0f (waiting for the stream %.0f) | triplets %d"
              % (tr[0] / n, tr[1] / n, tr[2] / n, tr[3] / n, tr[4] / n, tr[5] / n, tr[6] / n, tr[7] / n))
    print(("stream" if mode else "half  "), {n: round(ms[i] / max(cnt[i], 1), 4) for i, n in enumerate(_lib.KERNEL_CLASSES) if cnt[i]}, flush=True)
same = torch.equal(out[0][1], out[1][1]) and torch.equal(out[0][0], out[1][0])
print(os.environ.get("AG_LIB_PATH", "default"), material, n_obj, batch, "edges", int(csr.n_rel().sum()), "BITWISE", "EQUAL" if same else "DIFFERENT max %.3e" % (out[0][1] - out[1][1]).abs().max().item())
