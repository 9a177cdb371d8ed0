"""GPU box: ag_chamfer / ag_chamfer_masked of the library named by AG_LIB_PATH (default: in-tree) on a fixed set of random shapes (odd sizes, batched and shared
targets, scattered masks, NaN-free) -> an .npz of the results + the time of the 1 024 x 1 000 x 1 000 call; two runs with different libraries are compared
with --compare a.npz b.npz (bit equality).     python tools/chamfer_ab.py out.npz | python tools/chamfer_ab.py --compare a.npz b.npz"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    bad = [k for k in a.files if not np.array_equal(a[k].view(np.uint32) if a[k].dtype == np.float32 else a[k], b[k].view(np.uint32) if b[k].dtype == np.float32 else b[k])]
    print(f"{len(a.files)} arrays, {len(bad)} differ", bad)
    sys.exit(1 if bad else 0)
import torch
from adaptigraph_amd import losses
DEV = "cuda:0"
tg = lambda v: torch.from_numpy(np.ascontiguousarray(v)).to(DEV)
out = {}
for k, (B, N, M, batched) in enumerate([(1024, 1000, 1000, False), (7, 4096, 333, True), (3, 1, 1, False), (5, 37, 2050, True), (64, 1001, 999, False), (33, 5, 6400, True),
                                        (16, 2005, 2005, True)]):
    rng = np.random.default_rng(100 + k)
    x = rng.normal(0, 2, (B, N, 3)).astype(np.float32)
    y = rng.normal(0.3, 2, (B if batched else 1, M, 3)).astype(np.float32)
    out[f"plain_{k}"] = losses.chamfer(tg(x), tg(y)).cpu().numpy()
    if batched:
        xm = rng.random((B, N)) < 0.7; ym = rng.random((B, M)) < 0.6
        xm[:, 0] = True; ym[:, 0] = True
        out[f"masked_{k}"] = losses.mean_chamfer_device(tg(x), tg(y), tg(xm), tg(ym)).cpu().numpy()
rng = np.random.default_rng(1)
x, y = tg(rng.normal(0, 2, (1024, 1000, 3)).astype(np.float32)), tg(rng.normal(0.3, 2, (1, 1000, 3)).astype(np.float32))
for _ in range(3): losses.chamfer(x, y)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): losses.chamfer(x, y)
torch.cuda.synchronize()
print(f"{os.environ.get('AG_LIB_PATH', 'in-tree')}: chamfer 1024 x 1000 x 1000: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
np.savez(sys.argv[1], **out)
