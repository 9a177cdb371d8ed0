"""Synthetic particle clouds and graph inputs.

Pure numpy, deterministic, shared by tools/gen_golden.py, tests/ and bench.py so that
every leg (reference goldens, CPU oracle, HIP path) sees the same bytes.

Conventions follow the reference's rollout driver (src/planning/forward_dynamics.py:83-123):
y is "up", object particles occupy slots [0, n_p), tool/eef key-points the trailing slots,
attrs[:, 0] = 1 for objects and attrs[:, 1] = 1 for tools, p_instance = 1 (one instance).
"""
import numpy as np

MATERIALS = {
    # adj radius / top-k / connect_tools_all / #tool points : src/config/planning/{rope,granular,cloth}.yaml:10-14
    "rope": dict(radius=0.5, topk=10, connect_tools_all=False, n_tools=1),
    "granular": dict(radius=0.4, topk=20, connect_tools_all=False, n_tools=5),
    "cloth": dict(radius=0.75, topk=5, connect_tools_all=True, n_tools=1),
}


def rope_cloud(n, spacing, rng):
    """Polyline x = i*s, y = 0, z = 3 sin(2 pi i / n) + N(0, 0.01^2); one tool near the middle."""
    i = np.arange(n, dtype=np.float64)
    pts = np.stack([i * spacing, np.zeros(n), 3.0 * np.sin(2 * np.pi * i / n)], -1)
    pts = pts + rng.normal(0.0, 0.01, pts.shape)
    tool = np.array([[pts[n // 2, 0], 0.0, pts[n // 2, 2] + 0.3]])
    return pts.astype(np.float32), tool.astype(np.float32)


def granular_cloud(n, rng, density=80.0):
    """n points uniform in a thin square slab with ~40 points per radius-0.4 disc; 5 tools on a 1.0 bar."""
    side = float(np.sqrt(n / density))
    pts = np.stack([rng.uniform(0, side, n), rng.uniform(0, 0.05, n), rng.uniform(0, side, n)], -1)
    c = side / 2
    off = np.array([0.0, 0.5, 0.25, -0.25, -0.5])
    tool = np.stack([np.full(5, c), np.full(5, 0.0), c + off], -1)
    return pts.astype(np.float32), tool.astype(np.float32)


def cloth_cloud(side, rng, tool_near=True):
    """side x side grid, spacing 0.25 in x-z, + N(0, 0.01^2); 1 tool above the grid centre (or far away)."""
    g = np.arange(side, dtype=np.float64) * 0.25
    xx, zz = np.meshgrid(g, g, indexing="ij")
    pts = np.stack([xx.ravel(), np.zeros(side * side), zz.ravel()], -1)
    pts = pts + rng.normal(0.0, 0.01, pts.shape)
    c = g[-1] / 2
    tool = np.array([[c, 0.05, c]]) if tool_near else np.array([[c, 50.0, c]])
    return pts.astype(np.float32), tool.astype(np.float32)


def make_cloud(material, n_obj, rng, **kw):
    if material == "rope":
        return rope_cloud(n_obj, kw.get("spacing", 0.1), rng)
    if material == "granular":
        return granular_cloud(n_obj, rng)
    if material == "cloth":
        side = int(round(np.sqrt(n_obj)))
        assert side * side == n_obj, "cloth wants a square particle count"
        return cloth_cloud(side, rng, kw.get("tool_near", True))
    raise ValueError(material)


def make_graph_inputs(material, n_obj, batch, seed=0, n_his=4, n_pad=0, **kw):
    """Batched model inputs in the reference's dict layout (minus Rr/Rs; edges are built separately).

    n_pad extra *invalid* object slots (mask False, position 0) sit between the objects and the tools,
    reproducing the dataset padding contract ("padded node slots").
    Sample b > 0 is sample 0 plus a small per-sample perturbation so batches are not replicas.
    """
    rng = np.random.default_rng(seed)
    obj, tool = make_cloud(material, n_obj, rng, **kw)
    n_t = tool.shape[0]
    n_p = n_obj + n_pad
    N = n_p + n_t
    state = np.zeros((batch, n_his, N, 3), np.float32)
    for b in range(batch):
        cur = np.zeros((N, 3), np.float32)
        cur[:n_obj] = obj + rng.normal(0, 0.003, obj.shape).astype(np.float32) * (b > 0)
        cur[n_p:] = tool
        for h in range(n_his):
            jit = rng.normal(0, 0.01, (N, 3)).astype(np.float32)
            jit[n_obj:n_p] = 0
            state[b, h] = cur + jit
        state[b, :, n_obj:n_p] = 0
    attrs = np.zeros((batch, N, 2), np.float32)
    attrs[:, :n_obj, 0] = 1
    attrs[:, n_p:, 1] = 1
    action = np.zeros((batch, N, 3), np.float32)
    action[:, n_p:] = np.array([0.1, 0.0, 0.0], np.float32)
    p_instance = np.zeros((batch, n_p, 1), np.float32)
    p_instance[:, :n_obj] = 1
    phys = np.full((batch, 1), 0.5, np.float32)
    mask = np.zeros((batch, N), bool)
    mask[:, :n_obj] = True
    mask[:, n_p:] = True
    tool_mask = np.zeros((batch, N), bool)
    tool_mask[:, n_p:] = True
    return dict(state=state, attrs=attrs, action=action, p_instance=p_instance, phys=phys,
                mask=mask, tool_mask=tool_mask, n_p=n_p, n_obj=n_obj, n_tools=n_t)


def make_mpc_inputs(material, n_obj, bsz, n_look=1, seed=0, len_lo=2, len_hi=6, **kw):
    """state_cur (n_obj,3) + action sequences (bsz, n_look, 4) = [x, z, theta, length] for dynamics()."""
    rng = np.random.default_rng(seed)
    obj, _ = make_cloud(material, n_obj, rng, **kw)
    lo = obj.min(0)
    hi = obj.max(0)
    act = np.zeros((bsz, n_look, 4), np.float32)
    act[..., 0] = rng.uniform(lo[0], hi[0], (bsz, n_look))
    act[..., 1] = rng.uniform(lo[2] - 0.3, hi[2] + 0.3, (bsz, n_look))
    act[..., 2] = rng.uniform(-3.14, 3.14, (bsz, n_look))
    act[..., 3] = rng.uniform(len_lo, len_hi, (bsz, n_look))
    return obj, act
