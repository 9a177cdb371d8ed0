"""CPU study: what would a TWO-product MFMA scheme for the edge-encoder stack cost in accuracy?  float64 forward on the fwd_*
goldens with (A) the stack's weights rounded to fp16 (weights single-term, activations split) or (B) the stack's layer INPUTS
rounded to fp16 (activations single-term, weights split), layers 2..4 only or all four; max-abs pred_motion error vs the golden."""
import os, sys
import numpy as np, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, golden_files, weights_for
W0 = load_golden("weights_seed0")
W = {}
def use_weights(g):        # per golden: seed-0 default init, or the trained set the fixture names (tools/gen_trained.py)
    W.clear(); W.update({k: torch.from_numpy(v).double() for k, v in weights_for(g, W0).items()})
h16 = lambda x: x.half().double()
b16 = lambda x: x.bfloat16().double()


def forward(g, wq, xq, first, node=False, hsq=None, hrq=None, groups=None, etq=None, res_cols=0):
    use_weights(g)
    # groups: node-level layer groups whose INPUTS are rounded with xq: 'pe' particle_encoder, 'pp' particle_propagator (both
    # column blocks), 'rs' W_r / W_s (relation_propagator node blocks), 'dec' non_rigid_predictor
    G = lambda name: node or (groups is not None and name in groups)
    """wq / xq: rounding applied to the edge stack's weights / layer inputs (identity = exact); first: include layer 1."""
    t = lambda k: torch.from_numpy(np.ascontiguousarray(g[k])).double()
    state, attrs, action, p_inst, phys = t("state"), t("attrs"), t("action"), t("p_instance"), t("phys")
    B, N = attrs.shape[:2]; n_p = p_inst.shape[1]
    sn = torch.cat([state[:, 1:] - state[:, :-1], state[:, -1:]], 1).transpose(1, 2).reshape(B, N, -1)
    ph = torch.cat([phys[:, None].expand(B, n_p, -1), phys.new_zeros(B, N - n_p, phys.shape[1])], 1)
    p_in = torch.cat([attrs, ph, action], 2)
    grp = torch.cat([p_inst, p_inst.new_zeros(B, N - n_p, p_inst.shape[2])], 1)
    lin = lambda x, w, b, q: F.linear((xq if q else (lambda v: v))(x), (wq if q else (lambda v: v))(w), b)
    out = []
    for b in range(B):
        n = int(g["n_rel"][b]); r = torch.from_numpy(g["recv"][b, :n].astype("int64")); s = torch.from_numpy(g["send"][b, :n].astype("int64"))
        rel = torch.cat([attrs[b, r], attrs[b, s], (grp[b, r] - grp[b, s]).abs().sum(1, keepdim=True), sn[b, r] - sn[b, s]], 1)
        x = p_in[b]
        for i in (0, 2, 4): x = F.relu(lin(x, W[f"particle_encoder.model.{i}.weight"], W[f"particle_encoder.model.{i}.bias"], G('pe')))
        enc_n = x
        x = rel
        if res_cols and first:      # r03: the last inputs (x_r - x_s) carry their fp16 residual in spare K slots of layer 1 (ag_mlp.hip f16_residual)
            xe = xq(x); xe[:, -res_cols:] = xe[:, -res_cols:] + xq(x[:, -res_cols:] - xe[:, -res_cols:])
            x = F.relu(F.linear(xe, wq(W["relation_encoder.model.0.weight"]), W["relation_encoder.model.0.bias"]))
        else:
            x = F.relu(lin(x, W["relation_encoder.model.0.weight"], W["relation_encoder.model.0.bias"], first))
        for i in (2, 4): x = F.relu(lin(x, W[f"relation_encoder.model.{i}.weight"], W[f"relation_encoder.model.{i}.bias"], True))
        wrp, brp = W["relation_propagator.linear.weight"], W["relation_propagator.linear.bias"]
        wpp, bpp = W["particle_propagator.linear.weight"], W["particle_propagator.linear.bias"]
        eterm = lin(x, wrp[:, :150], brp, True)
        if etq is not None: eterm = etq(eterm)          # the per-edge table as stored (fp16 in precision mode 2)
        hcur = enc_n
        for _ in range(3):
            hr_t, hs_t = lin(hcur, wrp[:, 150:300], None, G('rs')), lin(hcur, wrp[:, 300:], None, G('rs'))
            if hsq is not None: hs_t = hsq(hs_t)       # what an fp16 copy of the gathered sender table would cost
            if hrq is not None: hr_t = hrq(hr_t)
            eff = F.relu(eterm + hr_t[r] + hs_t[s])
            agg = torch.zeros(N, 150, dtype=torch.float64).index_add_(0, r, eff)
            hcur = F.relu(lin(enc_n, wpp[:, :150], bpp, G('pp')) + lin(agg, wpp[:, 150:], None, G('pp')) + hcur)
        x = hcur[:n_p]
        x = F.relu(lin(x, W["non_rigid_predictor.linear_0.weight"], W["non_rigid_predictor.linear_0.bias"], G("dec")))
        x = F.relu(lin(x, W["non_rigid_predictor.linear_1.weight"], W["non_rigid_predictor.linear_1.bias"], G("dec")))
        out.append(lin(x, W["non_rigid_predictor.linear_2.weight"], W["non_rigid_predictor.linear_2.bias"], G("dec")))
    return torch.stack(out).numpy()


ident = lambda v: v
cases = [("exact f64", ident, ident, True, False),
         ("A: W fp16, layers 1-4", h16, ident, True, False), ("A: W fp16, layers 2-4", h16, ident, False, False),
         ("B: x fp16, layers 1-4", ident, h16, True, False), ("B: x fp16, layers 2-4", ident, h16, False, False),
         ("B: x fp16, edge 2-4 + all node layers", ident, h16, False, True),
         ("A: W fp16, edge 2-4 + all node layers", h16, ident, False, True),
         ("x bf16, layers 2-4 (one product)", ident, b16, False, False),
         ("C: W fp16 AND x fp16 (ONE product), 1-4", h16, h16, True, False)]
extra = [("Hs (gathered sender terms) rounded to fp16", dict(hsq=h16)), ("Hr (receiver terms) rounded to fp16", dict(hrq=h16)),
         ("B edge 1-4 + Eterm fp16 + Hs fp16", dict(hsq=h16))]
for label, kw in extra[:2]:
    errs = []
    for name in golden_files("fwd_"):
        g = load_golden(name)
        if float(g["decoder_scale"]) != 1.0: continue
        errs.append(f"{name[4:]} {np.abs(forward(g, ident, ident, True, False, **kw) - g['pred_motion']).max():.2e}")
    print(f"{label:40s}", " | ".join(errs))
for grp in ("pe", "pp", "rs", "dec"):
    errs = []
    for name in golden_files("fwd_"):
        g = load_golden(name)
        if float(g["decoder_scale"]) != 1.0: continue
        errs.append(f"{name[4:]} {np.abs(forward(g, ident, h16, True, False, groups={grp}) - g['pred_motion']).max():.2e}")
    print(f"{'B: x fp16, edge stack + node group ' + grp:40s}", " | ".join(errs))
for label, wq, xq, first, node in cases:
    errs = []
    for name in golden_files("fwd_"):
        g = load_golden(name)
        if float(g["decoder_scale"]) != 1.0: continue
        errs.append(f"{name[4:]} {np.abs(forward(g, wq, xq, first, node) - g['pred_motion']).max():.2e}")
    print(f"{label:40s}", " | ".join(errs))
# the shipped mode 2 and the one-product candidate, both WITH the fp16 per-edge table (what the engine would actually compute)
for label, wq, xq, res in (("mode 2 until r03: B 1-4 + Eterm fp16", ident, h16, 0), ("mode 2 as shipped: + residuals of x_r - x_s", ident, h16, 3),
                           ("ONE product: C 1-4 + Eterm fp16", h16, h16, 0)):
    errs = []
    for name in golden_files("fwd_"):
        g = load_golden(name)
        if float(g["decoder_scale"]) != 1.0: continue
        errs.append(f"{name[4:]} {np.abs(forward(g, wq, xq, True, False, etq=h16, res_cols=res) - g['pred_motion']).max():.2e}")
    print(f"{label:40s}", " | ".join(errs))
