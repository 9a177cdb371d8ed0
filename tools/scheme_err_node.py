"""CPU study (float64 emulation, no GPU), companion of tools/scheme_err.py: which NODE-level quantities of the default mode tolerate 16-bit storage or the edge
stack's H3 arithmetic.  Same random case stream as tools/fuzz_parity.py (tools/fuzz_cases.py); the edge stack runs the shipped H3 arithmetic with the q16 table.
    python tools/scheme_err_node.py [cases=80] [seed=11]
Result (74 cases, seed 11; worst max-abs deviation of the predicted motion from the exact forward):
    shipped (node stacks split-bf16, fp32 tables)                 8.3e-6
    + sender table Hs as q16 rows (rounds 1, 2)                   8.5e-6     <- shipped since r04: the errors of a receiver's senders are independent
    + receiver table Hr as q16 rows too                           1.5e-5        (Hr is common to all edges of a receiver: its error adds coherently)
    + Hs as plain fp16                                            1.9e-5
    + agg as q16 rows                                             1.8e-5
    propagator / decoder / all node layers on the H3 arithmetic   1.1e-5 / 1.6e-5 / 2.3e-5
r06: + agg as UNSIGNED q16 rows (shipped) 9.8e-6; on top of it Hr as q16 rows 1.6e-5 (one scale per 32) / 1.45e-5 (per 8), as 24-bit rows 9.7e-6."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools')); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import torch.nn.functional as F
import scheme_err as S
from scheme_err import *

def q_int_blk(x, blk, bits=16):      # q_int_tile with one scale per `blk` values
    Fdim = x.shape[-1]
    b = F.pad(x, (0, 160 - Fdim)).reshape(*x.shape[:-1], 160 // blk, blk)
    mx = b.abs().amax(-1, keepdim=True).clamp(min=1e-300)
    sc = torch.exp2(torch.floor(torch.log2(mx)) + 1)
    q = torch.round(b / sc * (2 ** (bits - 1) - 1)) / (2 ** (bits - 1) - 1) * sc
    return q.reshape(*x.shape[:-1], 160)[..., :Fdim]
def lin_h3(x, w, b=None):
    xe = h16(x); wh = h16(w)
    x8 = trunc_e5m2(xe)
    y = F.linear(xe, wh, b) + F.linear(x8, q_e4m3_rows(w - wh)) + F.linear(q_bf8(x - xe, 0), q_e4m3_rows(wh))
    return y
def lin_b3(x, w, b=None):      # split-bf16: both operands to 16 mantissa bits (hi + lo bf16), lo.lo dropped
    def sp(v):
        hi = v.float().bfloat16().double(); lo = (v - hi).float().bfloat16().double(); return hi, lo
    xh, xl = sp(x); wh, wl = sp(w)
    y = F.linear(xh, wh) + F.linear(xl, wh) + F.linear(xh, wl)
    return y if b is None else y + b

def forward2(W, g, n_rel, recv, send, sch, nlin, which):
    t = lambda k: torch.from_numpy(np.ascontiguousarray(g[k])).double()
    state, attrs, action, p_inst, phys = t("state"), t("attrs"), t("action"), t("p_instance"), t("phys")
    B, N = attrs.shape[:2]; n_p = p_inst.shape[1]
    sn = torch.cat([state[:, 1:] - state[:, :-1], state[:, -1:]], 1).transpose(1, 2).reshape(B, N, -1)
    ph = torch.cat([phys[:, None].expand(B, n_p, -1), phys.new_zeros(B, N - n_p, phys.shape[1])], 1)
    p_in = torch.cat([attrs, ph, action], 2)
    grp = torch.cat([p_inst, p_inst.new_zeros(B, N - n_p, p_inst.shape[2])], 1)
    L = lambda name: (nlin if name in which else (lambda x, w, b=None: F.linear(x, w, b)))
    out = []
    for b in range(B):
        n = int(n_rel[b]); r = torch.from_numpy(recv[b, :n].astype("int64")); s = torch.from_numpy(send[b, :n].astype("int64"))
        rel = torch.cat([attrs[b, r], attrs[b, s], (grp[b, r] - grp[b, s]).abs().sum(1, keepdim=True), sn[b, r] - sn[b, s]], 1)
        x = p_in[b]
        for i in (0, 2, 4): x = F.relu(F.linear(x, W[f"particle_encoder.model.{i}.weight"], W[f"particle_encoder.model.{i}.bias"]))
        enc_n = x
        x = rel
        for li, i in enumerate((0, 2, 4)):
            x = F.relu(sch.lin(x, W[f"relation_encoder.model.{i}.weight"], W[f"relation_encoder.model.{i}.bias"], li))
        wrp, brp = W["relation_propagator.linear.weight"], W["relation_propagator.linear.bias"]
        wpp, bpp = W["particle_propagator.linear.weight"], W["particle_propagator.linear.bias"]
        eterm = sch.table(sch.lin(x, wrp[:, :150], brp, 3))
        hcur = enc_n
        pn = F.linear(enc_n, wpp[:, :150], bpp)
        for rnd in range(3):
            if rnd == 0:
                hr_t, hs_t = F.linear(hcur, wrp[:, 150:300]), F.linear(hcur, wrp[:, 300:])     # from the node encoder kernel (split-bf16, once per rollout)
            else:
                hr_t, hs_t = L("hrhs")(hcur, wrp[:, 150:300]), L("hrhs")(hcur, wrp[:, 300:])
                if "hr_q16" in which: hr_t = q_int_tile(hr_t)
                if "hr_q16b8" in which: hr_t = q_int_blk(hr_t, 8)
                if "hr_q24" in which: hr_t = q_int_tile(hr_t, 24)
                if "hs_q16" in which: hs_t = q_int_tile(hs_t)
                if "hs_f16" in which: hs_t = h16(hs_t)
            eff = F.relu(eterm + hr_t[r] + hs_t[s])
            agg = torch.zeros(N, 150, dtype=torch.float64).index_add_(0, r, eff)
            if "agg_q16" in which: agg = q_int_tile(agg)
            if "agg_q16u" in which: agg = q_int_tile(agg, 17)      # unsigned 16 bits of a non-negative value = the magnitude bits of a signed 17-bit one
            hcur = F.relu(pn + L("prop")(agg, wpp[:, 150:]) + hcur)
        x = hcur[:n_p]
        x = F.relu(L("dec")(x, W["non_rigid_predictor.linear_0.weight"], W["non_rigid_predictor.linear_0.bias"]))
        x = F.relu(L("dec")(x, W["non_rigid_predictor.linear_1.weight"], W["non_rigid_predictor.linear_1.bias"]))
        out.append(L("dec2")(x, W["non_rigid_predictor.linear_2.weight"], W["non_rigid_predictor.linear_2.bias"]))
    return torch.stack(out).numpy()

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 80
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 11
edge = Scheme("H3X", "h3x", q_int_tile)
ALL = ("hrhs", "prop", "dec", "dec2")
variants = [("shipped: node stacks split-bf16, fp32 tables", lin_b3, ALL),
            ("+ Hs q16 (shipped since r04)", lin_b3, ALL + ("hs_q16",)), ("+ Hs q16 + Hr q16", lin_b3, ALL + ("hs_q16", "hr_q16")),
            ("+ Hs fp16", lin_b3, ALL + ("hs_f16",)), ("+ Hs q16 + agg q16", lin_b3, ALL + ("hs_q16", "agg_q16")),
            # r06 (profiles/r06_agg_q16.txt (4)): the shipped unsigned agg rows, and the receiver table on top of them
            ("+ Hs q16 + agg q16 UNSIGNED (shipped r06)", lin_b3, ALL + ("hs_q16", "agg_q16u")), ("shipped r06 + Hr q16 (per 32)", lin_b3, ALL + ("hs_q16", "agg_q16u", "hr_q16")),
            ("shipped r06 + Hr q16 (per 8)", lin_b3, ALL + ("hs_q16", "agg_q16u", "hr_q16b8")), ("shipped r06 + Hr q24", lin_b3, ALL + ("hs_q16", "agg_q16u", "hr_q24")),
            ("propagator on H3", lin_h3, ("prop",)), ("decoder layers 0, 1 on H3", lin_h3, ("dec",)), ("all node layers on H3", lin_h3, ALL)]
worst = {v[0]: (0.0, "") for v in variants}
n = 0
for case in gen_cases(cases, seed, 2, max_obj=1200):
    mat, g, wname, variant = case["mat"], case["g"], case["wname"], case["variant"]
    mm = synth.MATERIALS[mat]
    n_rel, recv, send = ago.build_edges(g["state"][:, -1], mm["radius"], g["mask"], g["tool_mask"], mm["topk"], mm["connect_tools_all"], variant)
    W = {k: torch.from_numpy(v).double() for k, v in WN[wname].items()}
    ref = S.forward(W, g, n_rel, recv, send, Scheme("exact", "exact"))
    mag = float(np.abs(ref).max()); line = []
    for name, nl, which in variants:
        e = float(np.abs(forward2(W, g, n_rel, recv, send, edge, nl, which) - ref).max()); line.append(e)
        if e > worst[name][0]: worst[name] = (e, case["tag"] + f" |motion| {mag:.3f}")
    n += 1
    print(f"{case['c']:4d} |m| {mag:.3f} " + " ".join(f"{e:.2e}" for e in line) + "  " + case["tag"].split(": ", 1)[1][:60], flush=True)
print("---", n, "cases")
for name, _, _ in variants: print(f"{name:44s} worst {worst[name][0]:.2e}  {worst[name][1]}")
