"""CPU study (float64 emulation on the trained + seed-0 forward goldens): max-abs pred_motion error when the inputs of ONE layer of the edge
stack (or a subset) are rounded to fp16, with / without the fp16 per-edge table -- which layer of precision mode 2 costs what.
    python tools/edge_layer_sensitivity.py      (appended to profiles/r03_two_product_err.txt)"""
import os, sys

import numpy as np, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT,'tools'))
from conftest import load_golden, golden_files, weights_for
W0 = load_golden("weights_seed0")
h16 = lambda x: x.half().double()
def forward(g, which, wq=False, etq=True):
    W = {k: torch.from_numpy(v).double() for k, v in weights_for(g, W0).items()}
    t = lambda k: torch.from_numpy(np.ascontiguousarray(g[k])).double()
    state, attrs, action, p_inst, phys = t("state"), t("attrs"), t("action"), t("p_instance"), t("phys")
    B, N = attrs.shape[:2]; n_p = p_inst.shape[1]
    sn = torch.cat([state[:, 1:] - state[:, :-1], state[:, -1:]], 1).transpose(1, 2).reshape(B, N, -1)
    ph = torch.cat([phys[:, None].expand(B, n_p, -1), phys.new_zeros(B, N - n_p, phys.shape[1])], 1)
    p_in = torch.cat([attrs, ph, action], 2)
    grp = torch.cat([p_inst, p_inst.new_zeros(B, N - n_p, p_inst.shape[2])], 1)
    out = []
    for b in range(B):
        n = int(g["n_rel"][b]); r = torch.from_numpy(g["recv"][b, :n].astype("int64")); s = torch.from_numpy(g["send"][b, :n].astype("int64"))
        rel = torch.cat([attrs[b, r], attrs[b, s], (grp[b, r] - grp[b, s]).abs().sum(1, keepdim=True), sn[b, r] - sn[b, s]], 1)
        x = p_in[b]
        for i in (0, 2, 4): x = F.relu(F.linear(x, W[f"particle_encoder.model.{i}.weight"], W[f"particle_encoder.model.{i}.bias"]))
        enc_n = x
        x = rel
        for li, i in enumerate((0, 2, 4)):
            xi = h16(x) if li in which else x
            x = F.relu(F.linear(xi, W[f"relation_encoder.model.{i}.weight"], W[f"relation_encoder.model.{i}.bias"]))
        wrp, brp = W["relation_propagator.linear.weight"], W["relation_propagator.linear.bias"]
        wpp, bpp = W["particle_propagator.linear.weight"], W["particle_propagator.linear.bias"]
        eterm = F.linear(h16(x) if 3 in which else x, wrp[:, :150], brp)
        if etq: eterm = h16(eterm)
        hcur = enc_n
        for _ in range(3):
            hr_t, hs_t = F.linear(hcur, wrp[:, 150:300]), F.linear(hcur, wrp[:, 300:])
            eff = F.relu(eterm + hr_t[r] + hs_t[s])
            agg = torch.zeros(N, 150, dtype=torch.float64).index_add_(0, r, eff)
            hcur = F.relu(F.linear(enc_n, wpp[:, :150], bpp) + F.linear(agg, wpp[:, 150:]) + hcur)
        x = hcur[:n_p]
        x = F.relu(F.linear(x, W["non_rigid_predictor.linear_0.weight"], W["non_rigid_predictor.linear_0.bias"]))
        x = F.relu(F.linear(x, W["non_rigid_predictor.linear_1.weight"], W["non_rigid_predictor.linear_1.bias"]))
        out.append(F.linear(x, W["non_rigid_predictor.linear_2.weight"], W["non_rigid_predictor.linear_2.bias"]))
    return torch.stack(out).numpy()
names=[n for n in golden_files("fwd_trained")] + ["fwd_granular205","fwd_rope64"]
for label, which, etq in (("Eterm fp16 only", (), True), ("x fp16 @L1", (0,), False), ("x fp16 @L2", (1,), False), ("x fp16 @L3", (2,), False), ("x fp16 @L4 (W_e)", (3,), False), ("x fp16 @L1-3 + Eterm", (0,1,2), True), ("x fp16 @L1,2,4 + Eterm", (0,1,3), True), ("x fp16 @L1,2 + Eterm",(0,1),True)):
    print(f"{label:26s}", " ".join(f"{n[4:].replace('trained_','t_')}:{np.abs(forward(load_golden(n), which, etq=etq) - load_golden(n)['pred_motion']).max():.1e}" for n in names), flush=True)
