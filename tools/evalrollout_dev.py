"""GPU: deviation of the eval-rollout error curves (the tests/test_eval_rollout.py fixture: 24-step rollouts with the graph
rebuilt from predicted positions every step) from the reference's, per engine precision mode and per rollout step."""
import os, sys, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_eval_rollout as T
from conftest import load_golden
from adaptigraph_amd import eval_rollout as er
w = load_golden("weights_seed0")
for prec in (0, 1, 2):
    with tempfile.TemporaryDirectory() as tmp:
        g = load_golden("evalrollout_rope")
        T.write_dataset(tmp, g)
        cfg = T.make_config(tmp, g)
        cfg["dataset_config"]["device"] = T.DEV
        model = T.engine_model(w, prec)
        out = os.path.join(tmp, "out"); os.makedirs(out)
        np.random.seed(int(g["seed"]))
        step_error = er.rollout_dataset(model, T.DEV, cfg, out)
        d = np.abs(step_error - g["error_short"])
        print(f"precision {prec}: error_short max dev {d.max():.2e}; per step:", " ".join(f"{x:.1e}" for x in d.reshape(-1)[:30]))
        for e in (1, 2):
            for k in (1, 2):
                dd = np.abs(np.loadtxt(os.path.join(out, str(e), "short", f"error_{k}.txt")) - g[f"error_{e}_{k}"])
                print(f"   episode {e} start {k}: max {dd.max():.2e}, first step outside 1e-4: {int(np.argmax(dd > 1e-4)) if (dd > 1e-4).any() else None} of {len(dd)}; values {g[f'error_{e}_{k}'][:3]}")
