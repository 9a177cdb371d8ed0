"""Debug: compare the q16 per-edge table written by the weight-stationary and the streaming edge encoder on one forward golden
(workspace layout mirrored from csrc/ag_api.hip carve_forward).   python tools/eterm_diff.py [golden]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from conftest import load_golden, weights_for
from adaptigraph_amd import configs, graph as aggraph
from adaptigraph_amd.model import DynamicsPredictor
from fwd_err import csr, t, DEV

name = sys.argv[1] if len(sys.argv) > 1 else "fwd_granular205"
g = load_golden(name); mat = str(g["material"])
w = weights_for(g, load_golden("weights_seed0"))
m = DynamicsPredictor(configs.model_config(), configs.material_config(mat), configs.dataset_config(mat), DEV)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w.items()}); m = m.to(DEV).eval()
c = csr(g["n_rel"], g["recv"], g["send"], g["attrs"].shape[1])
B, N = g["attrs"].shape[:2]
al = lambda v, a: (v + a - 1) // a * a
rows_pad, e_pad, rows_c = al(B * N, 128), al(max(c.e_cap, 1), 256), al(B * (N + 8), 128)
rc = rows_c + 128
off = 0
def take(nbytes):
    global off
    off = al(off, 256); r = off; off += nbytes; return r
for _ in range(6): take(rows_pad * 160 * 4)
for _ in range(4): take(rc * 160 * 4)
take(rows_pad * 4); take(rc * 4); take(rc * 4); take(e_pad * 4)
take(rows_pad * 160 * 4)
eterm_off = take(e_pad * 160 * 4)
E = int(c.row_ptr[-1].item())
kw = {"action": t(g["action"]), mat + "_physics_param": t(g["phys"])}
tabs = []
for ws_kernel in (0, 1):
    m.set_option("edge_stationary", ws_kernel)
    m(t(g["state"]), t(g["attrs"]), c, None, t(g["p_instance"]), **kw)
    torch.cuda.synchronize()
    buf = aggraph._WS[(torch.device(DEV).type, torch.device(DEV).index, torch.cuda.current_stream(torch.device(DEV)).cuda_stream)]
    tabs.append(buf[eterm_off:eterm_off + E * 320].cpu().numpy().reshape(E, 320).copy())
a, b = tabs
d = a != b
pad = np.zeros(320, bool); pad[285:288] = True; pad[316:320] = True      # bytes nobody writes
d[:, pad] = False
print(name, "E", E, "rows that differ", int(d.any(1).sum()), "bytes that differ", int(d.sum()))
rows = np.nonzero(d.any(1))[0]
for r in rows[:12]:
    cols = np.nonzero(d[r])[0]
    print(" row", r, "(block", r // 32, "lane", r % 32, ") byte offsets", cols[:24], "tiles", sorted(set((cols // 64).tolist())))
print("exponent bytes streaming", a[rows[:4], 280:285] if len(rows) else "", "ws", b[rows[:4], 280:285] if len(rows) else "")
for r in rows[:3]:
    for nm, tab in (("streaming", a), ("ws", b)):
        q = tab[r, 128:192].view(np.int16).astype(np.int32)
        print(nm, "row", r, "exp h0", tab[r, 280:285], "exp h1 (t0..3 @316.., t4 @312)", tab[r, 316:320], tab[r, 312], "tile 2 q: max|h0|", np.abs(q[:16]).max(), "max|h1|", np.abs(q[16:]).max(),
              "values h0", (q[:4] * 2.0 ** (int(tab[r, 282]) - 126) / 32767).round(6), "h1", (q[16:20] * 2.0 ** (int(tab[r, 282]) - 126) / 32767).round(6))
for nm, tab in (("streaming", a), ("ws", b)):
    q = np.abs(tab[:, :320].copy().view(np.int16).astype(np.int32).reshape(E, 5, 32))
    q[:, 4, 12:16] = 0; q[:, 4, 26:32] = 0; q[:, 4, 10:12] = 0          # padding halves of tile 4 (h = 0: q = 3; h = 1: q = 2 upper pair, q = 3)
    mx = q.max(2)
    too_big = (mx < 16383).sum()          # a block maximum below half of full scale: the exponent is one too large for the stored values
    sat = (mx >= 32767).sum()
    print(nm, "tiles whose max |q| < 16383 (exponent larger than the stored values need):", int(too_big), " tiles with a value at +-32767:", int(sat), "of", E * 5)
def decode(tab):
    q = tab[:, :320].copy().view(np.int16).astype(np.float64).reshape(E, 5, 32)
    eb = tab[:, 280:285].astype(np.int32)
    return q * (2.0 ** (eb - 126) / 32767.0)[:, :, None]
da, db = decode(a), decode(b)
dd = np.abs(da - db); dd[:, 4, 10:16] = 0; dd[:, 4, 26:32] = 0
quant = np.maximum(2.0 ** (a[:, 280:285].astype(np.int32) - 126), 2.0 ** (b[:, 280:285].astype(np.int32) - 126)) / 32767.0
big = dd > 1.01 * quant[:, :, None]
print("decoded values that differ by more than one quantum:", int(big.sum()), "largest difference / quantum", float((dd / quant[:, :, None]).max()))
idx = np.argwhere(dd > 0)
from collections import Counter
print("differing (tile, position) histogram:", Counter((int(t_), int(p_)) for _, t_, p_ in idx).most_common(8))
# which kernel is closer to the exact per-edge term where they differ by more than one quantum?
import torch.nn.functional as Fn
W64 = {k: torch.from_numpy(v).double() for k, v in w.items()}
st = torch.from_numpy(g["state"]).double(); at = torch.from_numpy(g["attrs"]).double(); pi = torch.from_numpy(g["p_instance"]).double()
n_p = pi.shape[1]
sn = torch.cat([st[:, 1:] - st[:, :-1], st[:, -1:]], 1).transpose(1, 2).reshape(B, N, -1)
grp = torch.cat([pi, pi.new_zeros(B, N - n_p, pi.shape[2])], 1)
ex = []
for bb in range(B):
    n = int(g["n_rel"][bb]); r = torch.from_numpy(g["recv"][bb, :n].astype("int64")); s_ = torch.from_numpy(g["send"][bb, :n].astype("int64"))
    x = torch.cat([at[bb, r], at[bb, s_], (grp[bb, r] - grp[bb, s_]).abs().sum(1, keepdim=True), sn[bb, r] - sn[bb, s_]], 1)
    for i in (0, 2, 4): x = Fn.relu(Fn.linear(x, W64[f"relation_encoder.model.{i}.weight"], W64[f"relation_encoder.model.{i}.bias"]))
    ex.append(Fn.linear(x, W64["relation_propagator.linear.weight"][:, :150], W64["relation_propagator.linear.bias"]))
ex = torch.cat(ex).numpy()
def to_feat(dec):        # (E, 5, 32) accumulator order -> (E, 160) feature order: position 16h + 4q + p of tile t is feature 32t + 8q + 4h + p
    out = np.zeros((E, 160))
    for h_ in range(2):
        for q_ in range(4):
            for p_ in range(4):
                out[:, np.arange(5) * 32 + 8 * q_ + 4 * h_ + p_] = dec[:, :, 16 * h_ + 4 * q_ + p_]
    return out
fa, fb = to_feat(da)[:, :150], to_feat(db)[:, :150]
print("max |table - exact|: streaming %.3e  ws %.3e   (max |exact| %.3f)" % (np.abs(fa - ex).max(), np.abs(fb - ex).max(), np.abs(ex).max()))
where = np.argwhere(np.abs(fa - fb) > 1e-12)
ea, eb_ = np.abs(fa - ex)[tuple(where.T)], np.abs(fb - ex)[tuple(where.T)]
print("on the %d entries where the two tables differ: mean |err| streaming %.3e ws %.3e; max %.3e %.3e" % (len(where), ea.mean(), eb_.mean(), ea.max(), eb_.max()))
for (e_, t_, p_) in np.argwhere(big)[:6]:
    f_ = 32 * t_ + 8 * ((p_ % 16) // 4) + 4 * (p_ // 16) + p_ % 4
    print("entry row", e_, "tile", t_, "pos", p_, "feature", f_, "streaming %.8f ws %.8f exact %.8f" % (da[e_, t_, p_], db[e_, t_, p_], ex[e_, f_]), "exp bytes", a[e_, 280:285], b[e_, 280:285],
          "recv/send", None)
