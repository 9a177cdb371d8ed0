"""The random case stream of tools/fuzz_parity.py (GPU sweep) and tools/scheme_err.py (float64 emulation of candidate arithmetics on the
SAME cases, CPU only): random material, particle count (1 .. 2500), batch, padded / randomly invalidated slots, per-sample physics
parameter, tool actions up to +-0.5, precision mode, node de-duplication on / off, seed-0 / trained weights, builder variant."""
import numpy as np
from adaptigraph_amd import synth


def gen_cases(cases, seed=0, only_prec=None, max_obj=None):
    rng = np.random.default_rng(seed)

    def loguniform(lo, hi):
        return int(round(np.exp(rng.uniform(np.log(lo), np.log(hi)))))

    for c in range(cases):
        mat = ("rope", "granular", "cloth")[rng.integers(3)]
        if mat == "cloth":
            side = loguniform(1, 45); n_obj = side * side
        else:
            n_obj = loguniform(1, 2500 if mat == "granular" else 1500)
        batch = int(rng.integers(1, 6))
        n_pad = int(rng.integers(0, 10)) if rng.random() < 0.5 else 0
        kw = dict(spacing=float(rng.choice([0.03, 0.1, 0.3]))) if mat == "rope" else (dict(tool_near=bool(rng.random() < 0.7)) if mat == "cloth" else {})
        g = synth.make_graph_inputs(mat, n_obj, batch, seed=int(rng.integers(1 << 30)), n_pad=n_pad, **kw)
        n_p, N = g["n_p"], g["attrs"].shape[1]
        if rng.random() < 0.4:                    # invalidate random object slots (a ragged cloud: mask False, no instance, no attribute)
            drop = rng.random((batch, n_obj)) < rng.uniform(0.02, 0.3)
            g["mask"][:, :n_obj] &= ~drop
            g["p_instance"][:, :n_obj, 0] *= ~drop
            g["attrs"][:, :n_obj, 0] *= ~drop
        if rng.random() < 0.5:                    # per-sample physics parameter and tool action
            g["phys"] = rng.uniform(0.0, 1.0, g["phys"].shape).astype(np.float32)
            amax = float(rng.choice([0.1, 0.2, 0.5]))
            g["action"][:, n_p:] = rng.uniform(-amax, amax, (batch, N - n_p, 3)).astype(np.float32)
        prec = int(rng.integers(3)) if only_prec is None else only_prec
        dedup = int(rng.choice([0, 2])); wname = str(rng.choice(["seed0", mat]))
        variant = "batch" if rng.random() < 0.8 else "single"
        tag = f"case {c}: {mat} n_obj {n_obj} batch {batch} pad {n_pad} {kw} prec {prec} dedup {dedup} weights {wname} {variant}"
        if max_obj is not None and n_obj > max_obj:
            continue
        yield dict(c=c, mat=mat, n_obj=n_obj, batch=batch, g=g, prec=prec, dedup=dedup, wname=wname, variant=variant, tag=tag)
