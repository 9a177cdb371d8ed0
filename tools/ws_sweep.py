"""GPU: randomised shapes, weight-stationary vs streaming two-product edge encoder, bitwise (forward outputs) — covers the block-count /
workgroup-count / pipeline-depth corner cases (n_i = 1, 2, ..., 11 blocks per workgroup, partial last blocks, single-graph batches)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T
from adaptigraph_amd import synth
from adaptigraph_amd import graph as aggraph
w = T.load_golden("weights_seed0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
models = {m: T.make_model(w, m, prec="fast") for m in ("rope", "granular", "cloth")}
bad = 0
cases = [("rope", 33, 1), ("rope", 7, 1), ("rope", 1000, 27), ("rope", 1000, 28), ("rope", 1000, 30), ("rope", 1000, 55)]   # ~256 x k blocks
for _ in range(40):
    mat = ["rope", "granular", "cloth"][int(rng.integers(3))]
    n = int(rng.integers(5, 1200)) if mat != "cloth" else int(rng.integers(2, 30)) ** 2
    cases.append((mat, n, int(rng.integers(1, 9))))
for mat, n, B in cases:
    m = models[mat]
    kw = dict(spacing=0.1) if mat == "rope" else {}
    g = synth.make_graph_inputs(mat, n, B, seed=int(rng.integers(1 << 30)), **kw)
    mm = synth.MATERIALS[mat]
    csr = aggraph.build_edges(T.t(g["state"][:, -1]), mm["radius"], T.t(g["mask"]), T.t(g["tool_mask"]), mm["topk"], mm["connect_tools_all"],
                              "batch", max_tools=g["n_tools"])
    args = (T.t(g["state"]), T.t(g["attrs"]), csr, None, T.t(g["p_instance"]))
    kw2 = {"action": T.t(g["action"]), mat + "_physics_param": T.t(g["phys"])}
    m.set_option("edge_stationary", 0); _, a = m(*args, **kw2)
    m.set_option("edge_stationary", 1); _, b = m(*args, **kw2)
    E = int(csr.row_ptr[-1].item())
    ok = torch.equal(a, b) and bool(torch.isfinite(b).all()) and m.take_status() == 0
    bad += not ok
    print(f"{mat:9s} n {n:5d} B {B:3d} E {E:8d} blocks {(E + 31) // 32:7d}  {'ok' if ok else 'MISMATCH'}")
print("MISMATCHES:", bad)
