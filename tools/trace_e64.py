"""Phase trace of one wave's row tile in edge_encode_nb_kernel (debug build -DAG_TRACE=1): (tag, s_memtime) pairs.
   AG_LIB_PATH=ab/libtrace.so python tools/trace_e64.py
Tags: 10 row-tile top | 11 first layer done | 12 claim published | 13/14 RE1/RE2 done | 15 gathers issued | 16 We done |
17 features done | 18 next tile read;  per weight tile: 1 top | 2 k-loop issued | 3 fragments settled | 4 DMA drained | 5 barrier passed."""
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
for s in range(4):
    st = a[s]; n = int((st[:, 0] != 0).sum()); st = st[:n]
    if n == 0: continue
    t0 = st[0, 1]
    print(f"slot {s}: {n} stamps, row tile total {st[-1, 1] - t0} ticks")
    line = []
    prev = t0
    for tag, tm in st:
        line.append(f"{tag}:{tm - prev}")
        prev = tm
    print("   " + " ".join(line))
    # per weight tile breakdown
    seg = {1: [], 2: [], 3: [], 4: [], 5: []}
    for (tg0, t_0), (tg1, t_1) in zip(st[:-1], st[1:]):
        if tg1 in seg: seg[tg1].append(t_1 - t_0)
    print("   mean per weight tile: to-top %.0f  k-loop %.0f  settle %.0f  drain %.0f  barrier %.0f" % tuple(np.mean(seg[k]) if seg[k] else 0 for k in (1, 2, 3, 4, 5)))
