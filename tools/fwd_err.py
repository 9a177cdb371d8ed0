"""Max-abs deviation of ag_forward from the reference goldens (tests/golden/fwd_*.npz) per engine precision mode.
   [AG_LIB_PATH=...] python tools/fwd_err.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, golden_files, weights_for
from adaptigraph_amd import configs
from adaptigraph_amd.graph import CSREdges
from adaptigraph_amd.model import DynamicsPredictor
DEV = "cuda:0"
w0 = load_golden("weights_seed0")
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def csr(n_rel, recv, send, N):
    B = len(n_rel)
    r = np.concatenate([recv[b, :n] + b * N for b, n in enumerate(n_rel)]).astype(np.int32)
    s = np.concatenate([send[b, :n] + b * N for b, n in enumerate(n_rel)]).astype(np.int32)
    row_ptr = np.zeros(B * N + 1, np.int32)
    np.add.at(row_ptr, r + 1, 1)
    row_ptr = np.cumsum(row_ptr).astype(np.int32)
    pad = lambda a: np.concatenate([a, np.zeros(1, np.int32)]) if len(a) == 0 else a
    return CSREdges(t(row_ptr), t(pad(r)), t(pad(s)), B, N, len(r))


if __name__ == "__main__":
  for prec in (0, 1, 2):
    worst = 0.0
    for name in golden_files("fwd_"):
        g = load_golden(name)
        mat = str(g["material"]) if "material" in g else "rope"
        m = DynamicsPredictor(configs.model_config(), configs.material_config(mat), configs.dataset_config(mat), DEV)
        sd = {k: torch.from_numpy(v.copy()) for k, v in weights_for(g, w0).items()}    # seed-0 init or the trained set the golden names
        for k in sd:                                                  # the clamp golden scales the decoder (tools/gen_golden.py)
            if k.startswith("non_rigid_predictor.linear_2"): sd[k] *= float(g["decoder_scale"])
        m.load_state_dict(sd); m = m.to(DEV).eval(); m.set_option("precision", prec)
        pos, mot = m(t(g["state"]), t(g["attrs"]), csr(g["n_rel"], g["recv"], g["send"], g["attrs"].shape[1]), None, t(g["p_instance"]),
                     action=t(g["action"]), **{mat + "_physics_param": t(g["phys"])})
        scale = max(1.0, float(np.abs(g["pred_motion"]).max()))
        e = float(np.abs(mot.cpu().numpy() - g["pred_motion"]).max())
        worst = max(worst, e / scale)
        st = m.take_status()
        print(f"prec {prec} {name:34s} status {st} max-abs {e:.3e}  (|motion| max {np.abs(g['pred_motion']).max():.3f}, scaled {e / scale:.3e})")
    print(f"prec {prec} worst max-abs / max(1, |motion| max) {worst:.3e}")
