"""Phase trace of one wave's weight-tile loop in edge_encode_kernel<PrecB3> (debug build -DAG_TRACE=1).
   AG_LIB_PATH=ab/libtrace.so python tools/trace_tiles.py
Stamps (s_memtime, 100 MHz-independent shader clock ticks): per tile  top | dma issued | prologue+deferred epilogue |
k-loop issued | dma drained | barrier passed.  Rows: blocks 0, 1, 256, 257, wave 0, 6th row tile."""
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
a = np.array(buf, dtype=np.uint64).reshape(4, 512).astype(np.int64)
names = ["dma", "epi", "kloop", "drain", "barrier", "next"]
for s in range(4):
    st = a[s]
    n = int((st != 0).sum())
    st = st[:n]
    print(f"slot {s}: {n} stamps, row tile total {st[-1] - st[0]} ticks; first-layer+gather {st[1] - st[0]}")
    body = st[1:1 + ((n - 1) // 6) * 6].reshape(-1, 6)
    d = np.diff(np.concatenate([body.reshape(-1), st[1 + body.size:1 + body.size + 1] if n > 1 + body.size else body.reshape(-1)[-1:]]))
    d = d[: body.shape[0] * 6 - (0 if n > 1 + body.size else 1)]
    pad = np.zeros(body.shape[0] * 6, np.int64); pad[:len(d)] = d
    tbl = pad.reshape(-1, 6)
    print("   tile  " + " ".join(f"{x:>8s}" for x in names))
    for i, r in enumerate(tbl):
        print(f"   {i:4d}  " + " ".join(f"{x:8d}" for x in r))
    print("   mean  " + " ".join(f"{x:8.0f}" for x in tbl[:-1].mean(0)), " sum/tile", tbl[:-1].sum(1).mean())
