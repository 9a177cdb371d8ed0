"""Key-point down-sampling used to turn a recorded particle cloud into graph nodes — SURVEY.md §8f row n3.

`fps` mirrors src/dynamics/dataset/graph.py:8-36: a farthest-point pass down to `max_nobj` points followed by a
radius-limited farthest-point pass (`fps_rad_idx`, src/dynamics/utils.py:10-24).  The first pass is
`dgl.geometry.farthest_point_sampler` in the reference (DGL 1.x/2.x, not vendored and not installed here); its published
algorithm is restated in `farthest_point_sampler` below: start at `start_idx`, keep for every point the squared distance
to the nearest picked point, pick the arg-max (first index on ties), repeat.  Parity of that stage is therefore pinned to
the algorithm, not to DGL outputs.  Both passes draw from numpy's global RNG in the reference's order (start index of pass 1,
optional radius draw, start index of pass 2), so `np.random.seed(s)` reproduces the reference's node sets.

Host code on purpose: this is per-episode data preparation over a few thousand points, not the rollout path.
"""
import numpy as np


def farthest_point_sampler(pos, npoints, start_idx=None):
    """pos (B,N,3) array-like -> (B,npoints) int64 indices, DGL semantics (squared distances, first arg-max)."""
    pos = np.asarray(pos, np.float32)
    B, N, _ = pos.shape
    assert 0 < npoints <= N
    out = np.zeros((B, npoints), np.int64)
    for b in range(B):
        cur = np.random.randint(0, N) if start_idx is None else int(start_idx)
        near = np.full(N, np.inf, np.float32)
        for k in range(npoints):
            out[b, k] = cur
            d = pos[b] - pos[b, cur]
            near = np.minimum(near, (d * d).sum(1, dtype=np.float32))
            cur = int(near.argmax())
    return out


def fps_rad_idx(pcd, radius):
    """Farthest-point picks until every point is within `radius` of a pick -> (picked points, their indices)."""
    first = np.random.randint(pcd.shape[0])
    picks = [first]
    near = np.linalg.norm(pcd - pcd[first], axis=1)
    while near.max() > radius:
        nxt = near.argmax()
        picks.append(nxt)
        near = np.minimum(near, np.linalg.norm(pcd - pcd[nxt], axis=1))
    picks = np.stack(picks, axis=0)
    return pcd[picks], picks


def fps(obj_kp_start, max_nobj, fps_radius_range, verbose=False):
    """obj_kp_start (N,3) -> indices (n_fps,) into it; `fps_radius_range` is a float or a [lo, hi] range to draw from."""
    n = obj_kp_start.shape[0]
    coarse = farthest_point_sampler(obj_kp_start[None].astype(np.float32), min(max_nobj, n),
                                    start_idx=np.random.randint(0, n))[0].astype(np.int32)
    if type(fps_radius_range) == float:
        radius = fps_radius_range
    elif len(fps_radius_range) == 2:
        radius = np.random.uniform(fps_radius_range[0], fps_radius_range[1])
    else:
        raise ValueError(f"Invalid fps_radius_range: {fps_radius_range}.")
    _, fine = fps_rad_idx(obj_kp_start[coarse].astype(np.float32), radius)
    idx = coarse[fine.astype(np.int32)]
    if verbose:
        print(f"FPS num particles: {len(idx)} with index list \n {idx}. \n")
    return np.array(idx)
