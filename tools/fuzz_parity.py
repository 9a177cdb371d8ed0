"""Randomised whole-path parity sweep: edge builder (bit-exact) and one forward (1e-4 gate) against the CPU oracle over random materials, particle
counts (1 .. 2500), batches, padded / randomly invalidated slots, per-sample physics parameters and tool actions, precision modes, node
de-duplication on / off and seed-0 / trained weights.  Prints the worst deviation per precision mode and every failure; exit status 1 on any.
Every mode is held to the absolute gate at any motion size and must leave the model status at 0.
    python tools/fuzz_parity.py [cases=120] [seed=0] [precision: 0 / 1 / 2, default random per case]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden
from adaptigraph_amd import configs, synth
from adaptigraph_amd import graph as aggraph
from adaptigraph_amd.model import DynamicsPredictor
from oracle import ag_oracle as ago
from fuzz_cases import gen_cases

DEV = "cuda:0"
GATE = 1e-4
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
only_prec = int(sys.argv[3]) if len(sys.argv) > 3 else None
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
W = {"seed0": load_golden("weights_seed0")}
for mat in ("rope", "granular", "cloth"):
    W[mat] = load_golden("weights_trained_" + mat)
models = {}


def model(mat, wname, prec, dedup):
    key = (mat, wname)
    if key not in models:
        m = DynamicsPredictor(configs.model_config(), configs.material_config(mat), configs.dataset_config(mat), DEV)
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in W[wname].items()})
        models[key] = m.to(DEV).eval()
    m = models[key]
    m.set_option("precision", prec)
    m.set_option("node_dedup", dedup)
    return m


worst = {0: 0.0, 1: 0.0, 2: 0.0}
worst_rel = {0: 0.0, 1: 0.0, 2: 0.0}      # deviation / max(|reference motion|) of the case
fails = 0
max_mag = 0.0
for case in gen_cases(cases, seed, only_prec):
    mat, g, prec, dedup, wname, variant, tag = (case[k] for k in ("mat", "g", "prec", "dedup", "wname", "variant", "tag"))
    mm = synth.MATERIALS[mat]
    pos_now = g["state"][:, -1]
    n_rel, recv, send = ago.build_edges(pos_now, mm["radius"], g["mask"], g["tool_mask"], mm["topk"], mm["connect_tools_all"], variant)
    csr = aggraph.build_edges(t(pos_now), mm["radius"], t(g["mask"]), t(g["tool_mask"]), mm["topk"], mm["connect_tools_all"], variant,
                              max_tools=g["n_tools"])
    ok = csr.n_rel().cpu().tolist() == n_rel.tolist()
    if ok:
        for b, (r, s) in enumerate(csr.to_lists()):
            ok = ok and np.array_equal(r, recv[b, :n_rel[b]]) and np.array_equal(s, send[b, :n_rel[b]])
    if not ok:
        fails += 1; print("EDGE MISMATCH", tag); continue
    ref_pos, ref_mot = ago.forward(W[wname], g["state"], g["attrs"], g["action"], g["p_instance"], g["phys"], n_rel, recv, send)
    m = model(mat, wname, prec, dedup)
    pos, mot = m(t(g["state"]), t(g["attrs"]), csr, None, t(g["p_instance"]), action=t(g["action"]), **{mat + "_physics_param": t(g["phys"])})
    e = float(np.abs(mot.cpu().numpy() - ref_mot).max()); ep = float(np.abs(pos.cpu().numpy() - ref_pos).max())
    st = m.take_status()
    worst[prec] = max(worst[prec], e)
    mag = float(np.abs(ref_mot).max()); max_mag = max(max_mag, mag)
    worst_rel[prec] = max(worst_rel[prec], e / max(mag, 1e-3))
    if os.environ.get("FUZZ_VERBOSE"): print(f"{e:.3e} |motion| {mag:.3f} rel {e / max(mag, 1e-3):.2e}", tag)
    if not (e <= GATE and ep <= GATE and st == 0):
        fails += 1; print(f"FORWARD {e:.3e} / {ep:.3e} status {st} (max |reference motion| {mag:.3f})", tag)
print(f"{cases} cases, {fails} failures; largest reference motion {max_mag:.3f}; worst motion deviation per precision mode (f32, bf16x3, fast): "
      f"{worst[0]:.2e} {worst[1]:.2e} {worst[2]:.2e}; relative to the case's max |motion|: {worst_rel[0]:.2e} {worst_rel[1]:.2e} {worst_rel[2]:.2e}")
sys.exit(1 if fails else 0)
