"""Localise a precision-mode-2 deviation: one forward golden under every kernel choice of that mode.
    python tools/mode2_diag.py [golden=fwd_trained_rope_rope64]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, weights_for
from adaptigraph_amd import configs
from adaptigraph_amd.model import DynamicsPredictor
sys.path.insert(0, os.path.join(ROOT, "tools"))
from fwd_err import csr, t, DEV          # noqa: E402  (prints its table on import: fine)

name = sys.argv[1] if len(sys.argv) > 1 else "fwd_trained_rope_rope64"
g = load_golden(name); mat = str(g["material"])
w = weights_for(g, load_golden("weights_seed0"))
m = DynamicsPredictor(configs.model_config(), configs.material_config(mat), configs.dataset_config(mat), DEV)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w.items()}); m = m.to(DEV).eval()
c = csr(g["n_rel"], g["recv"], g["send"], g["attrs"].shape[1])
kw = {"action": t(g["action"]), mat + "_physics_param": t(g["phys"])}
def run(label, **opts):
    for k, v in opts.items(): m.set_option(k, v)
    _, mot = m(t(g["state"]), t(g["attrs"]), c, None, t(g["p_instance"]), **kw)
    e = float(np.abs(mot.cpu().numpy() - g["pred_motion"]).max())
    print(f"{label:70s} {e:.3e} status {m.take_status()}")
    return mot
print("---", name)
run("precision 1 (split-bf16, fp32 table)", precision=1)
a = run("precision 2, edge_products 3 (split-bf16 stack, q16 table)", precision=2, edge_products=3, edge_stationary=1, fuse_aggregate=0)
run("  + fuse_aggregate 2", fuse_aggregate=2)
b = run("precision 2, edge_products 2, streaming kernel", edge_products=2, edge_stationary=0, fuse_aggregate=0)
c2 = run("precision 2, edge_products 2, weight-stationary kernel", edge_stationary=1)
print("ws == streaming bitwise:", bool(torch.equal(b, c2)), " max diff", float((b - c2).abs().max()))
run("  node_dedup 0", node_dedup=0)
run("  node_dedup 2", node_dedup=2)
