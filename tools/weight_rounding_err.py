"""What would storing the WEIGHTS in fp16 (or bf16) cost?  Exact-fp32 engine with rounded weights vs the reference goldens
(unrounded weights): isolates the weight-quantisation error of a hypothetical single-term-weight MFMA mode."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, golden_files
from adaptigraph_amd import configs
from adaptigraph_amd.graph import CSREdges
from adaptigraph_amd.model import DynamicsPredictor
DEV = "cuda:0"
w = load_golden("weights_seed0")
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


EDGE_ONLY = len(sys.argv) > 1 and sys.argv[1] == "edge"       # round only the edge-encoder stack (relation_encoder + W_rp[:, :150])


def cast_weights(cast):
    out = {}
    for k, v in w.items():
        v = torch.from_numpy(v)
        if not EDGE_ONLY or k.endswith("bias"):
            out[k] = cast(v) if not EDGE_ONLY else v
        elif k.startswith("relation_encoder"):
            out[k] = cast(v)
        elif k == "relation_propagator.linear.weight":
            out[k] = torch.cat([cast(v[:, :150]), v[:, 150:]], 1)
        else:
            out[k] = v
    return out


for label, cast in (("exact", lambda x: x), ("fp16 weights", lambda x: x.half().float()), ("bf16 weights", lambda x: x.bfloat16().float())):
    for name in golden_files("fwd_"):
        g = load_golden(name)
        if float(g["decoder_scale"]) != 1.0:
            continue
        mat = str(g["material"])
        m = DynamicsPredictor(configs.model_config(), configs.material_config(mat), configs.dataset_config(mat), DEV)
        m.load_state_dict(cast_weights(cast)); m = m.to(DEV).eval(); m.set_option("precision", 0)
        pos, mot = m(t(g["state"]), t(g["attrs"]), csr(g["n_rel"], g["recv"], g["send"], g["attrs"].shape[1]), None, t(g["p_instance"]),
                     action=t(g["action"]), **{mat + "_physics_param": t(g["phys"])})
        print(f"{label:13s} {name:18s} max-abs {np.abs(mot.cpu().numpy() - g['pred_motion']).max():.3e}")
