"""Open-loop evaluation rollouts over a recorded dataset on the engine — SURVEY.md §8f row n3.

Mirrors src/dynamics/rollout/rollout.py (:20-319, minus image/video output) and the graph set-up of
src/dynamics/rollout/graph.py:233-399.  The reference rolls ONE start graph at a time, rebuilding dense Rr/Rs on the host
and copying state back and forth every step.  Which frames a rollout visits depends only on the frame-pair table, never on
the predictions, so here the frame schedule of every start graph is resolved on the host up front (`frame_schedule`), the
tool trajectories and ground-truth key-points are uploaded once, and ALL start graphs advance together as one batch:
per step one `ag_build_edges` + one `ag_forward` for the whole batch, the error reduction on device, one copy back at the
end (`rollout_batch`).  `rollout_from_start_graph` keeps the reference's per-graph signature on top of it.
"""
import glob
import os

import numpy as np
import torch

from .graph import build_edges, construct_edges_from_states
from .load import load_dataset, load_positions
from .sampling import fps

ROLLOUT_STEPS = 100      # rollout.py:62


def pad(x, max_dim, dim=0):
    """Zero-pad axis `dim` (0 of a 2-D, 1 of a 3-D array) to `max_dim`, fp32   (src/dynamics/utils.py:26-35)."""
    shape = list(x.shape)
    shape[dim] = max_dim
    out = np.zeros(shape, np.float32)
    out[tuple(slice(0, s) for s in x.shape)] = x
    return out


def pad_torch(x, max_dim, dim=0):
    shape = list(x.shape)
    shape[dim] = max_dim
    out = torch.zeros(shape, dtype=x.dtype, device=x.device)
    out[tuple(slice(0, s) for s in x.shape)] = x
    return out


def _dataset_params(dataset_config):
    d = dataset_config["datasets"][0]
    return dict(max_nobj=d["max_nobj"], max_nR=d["max_nR"], fps_radius=(d["fps_radius_range"][0] + d["fps_radius_range"][1]) / 2,
                adj_thresh=(d["adj_radius_range"][0] + d["adj_radius_range"][1]) / 2, topk=d["topk"],
                connect_tool_all=d["connect_tool_all"])


def start_graph_arrays(dataset_config, material_config, eef_pos, obj_pos, n_his, pair):
    """Host half of construct_graph: numpy arrays of everything but the edges.  -> (dict, fps_idx_list)."""
    P = _dataset_params(dataset_config)
    max_nobj, n_eef = P["max_nobj"], eef_pos.shape[1]
    n_state = max_nobj + n_eef
    frames = np.asarray(pair)
    obj_kps, eef_kps = np.asarray(obj_pos)[frames], np.asarray(eef_pos)[frames]
    fps_idx_list = fps(obj_kps[n_his - 1], max_nobj, P["fps_radius"])
    n_kp = len(fps_idx_list)

    state_history = np.zeros((n_his, n_state, obj_kps.shape[-1]), np.float32)
    state_history[:, :n_kp] = obj_kps[:n_his][:, fps_idx_list]
    state_history[:, max_nobj:] = eef_kps[:n_his]
    eef_kp = np.stack([eef_kps[n_his - 1], eef_kps[n_his]], axis=0).astype(np.float32)
    states_delta = np.zeros((n_state, obj_kps.shape[-1]), np.float32)
    states_delta[max_nobj:] = eef_kp[1] - eef_kp[0]

    state_mask = np.zeros(n_state, bool)
    state_mask[:n_kp] = True
    state_mask[max_nobj:] = True
    eef_mask = np.zeros(n_state, bool)
    eef_mask[max_nobj:] = True
    attrs = np.zeros((n_state, 2), np.float32)
    attrs[:n_kp, 0] = 1.0
    attrs[max_nobj:, 1] = 1.0
    p_instance = np.zeros((max_nobj, 1), np.float32)
    p_instance[:n_kp, 0] = 1
    assert len(dataset_config["materials"]) == 1, "only support single material"
    material_idx = np.zeros((max_nobj, len(material_config["material_index"])), np.int64)
    material_idx[:n_kp, material_config["material_index"][dataset_config["materials"][0]]] = 1
    arrays = {"state": state_history, "action": states_delta, "attrs": attrs, "p_rigid": np.zeros(1, np.float32),
              "p_instance": p_instance, "state_mask": state_mask, "eef_mask": eef_mask, "obj_mask": np.arange(max_nobj) < n_kp,
              "material_index": material_idx, "eef_kp": eef_kp}
    return arrays, fps_idx_list


def construct_graph(dataset_config, material_config, eef_pos, obj_pos, n_his, pair, physics_param, device="cuda"):
    """Start graph of one frame pair (graph.py:233-371): FPS key-points of frame pair[n_his-1], history from pair[:n_his],
    tool motion pair[n_his-1] -> pair[n_his], everything padded to max_nobj / max_nR, tensors on `device`.
    -> (graph dict with the reference's keys, fps_idx_list)."""
    P = _dataset_params(dataset_config)
    arrays, fps_idx_list = start_graph_arrays(dataset_config, material_config, eef_pos, obj_pos, n_his, pair)
    dev = torch.device(device)
    graph = {k: torch.from_numpy(v).to(dev) for k, v in arrays.items()}
    Rr, Rs = construct_edges_from_states(graph["state"][-1], P["adj_thresh"], graph["state_mask"], graph["eef_mask"], P["topk"],
                                         P["connect_tool_all"])
    graph["Rr"], graph["Rs"] = pad_torch(Rr, P["max_nR"]), pad_torch(Rs, P["max_nR"])
    for name, v in physics_param.items():
        graph[name + "_physics_param"] = torch.from_numpy(np.asarray(v, np.float32)).to(dev)
    return graph, fps_idx_list


def _middle_successor(pairs, n_his, current_end):
    nxt = pairs[(pairs[:, n_his - 1] == current_end) & (pairs[:, n_his] > current_end)]
    return nxt[len(nxt) // 2] if len(nxt) else None


def get_next_pair_or_break_episode_pushes(pairs, n_his, n_frames, current_end):
    """Continue from the frame the last prediction ended on: the middle one of the pairs starting there (graph.py:392-399)."""
    return _middle_successor(pairs, n_his, current_end)


def get_next_pair_or_break_episode(pairs, n_his, n_frames, current_end):
    """Same, but skip forward over frames no pair starts at (graph.py:373-390)."""
    nxt = _middle_successor(pairs, n_his, current_end)
    while nxt is None and current_end < n_frames:
        current_end += 1
        nxt = _middle_successor(pairs, n_his, current_end)
    return nxt


def frame_schedule(pairs, n_his, n_frames, current_start, current_end, next_fn, rollout_steps=ROLLOUT_STEPS):
    """[(start, end)] of every model step a rollout from (current_start, current_end) takes (rollout.py:66-93)."""
    sched = [(int(current_start), int(current_end))]
    while len(sched) < rollout_steps:
        nxt = next_fn(pairs, n_his, n_frames, sched[-1][1])
        if nxt is None:
            break
        sched.append((int(nxt[n_his - 1]), int(nxt[n_his])))
    return sched


@torch.no_grad()
def rollout_batch(model, device, graphs, fps_idx_lists, schedules, eef_pos_list, obj_pos_list, dataset_config):
    """Advance B start graphs together.  graphs[b] from construct_graph, schedules[b] from frame_schedule, eef_pos_list[b]
    (T,N_eef,3) / obj_pos_list[b] (T,N_obj_all,3) the episode the graph came from.  -> [error list per graph]; entry t is
    the mean key-point distance to the ground truth of frame schedules[b][t][1]."""
    P = _dataset_params(dataset_config)
    dev = torch.device(device)
    B, max_nobj = len(graphs), P["max_nobj"]
    n_eef = eef_pos_list[0].shape[1]
    T = max(len(s) for s in schedules)
    gt = np.zeros((B, T, max_nobj, 3), np.float32)
    eef_start = np.zeros((B, T, n_eef, 3), np.float32)
    eef_delta = np.zeros((B, T, n_eef, 3), np.float32)
    for b, sched in enumerate(schedules):
        for t, (s, e) in enumerate(sched):
            kp = np.asarray(obj_pos_list[b][e])[fps_idx_lists[b]]
            gt[b, t, :len(kp)] = kp
            eef_start[b, t] = eef_pos_list[b][s]
            eef_delta[b, t] = np.asarray(eef_pos_list[b][e]) - np.asarray(eef_pos_list[b][s])
    gt, eef_start, eef_delta = (torch.from_numpy(a).to(dev) for a in (gt, eef_start, eef_delta))
    stack = lambda k: torch.stack([g[k].to(dev) for g in graphs], 0)
    state, action, attrs, p_instance = stack("state").float(), stack("action").float(), stack("attrs"), stack("p_instance")
    state_mask, eef_mask, obj_mask = stack("state_mask"), stack("eef_mask"), stack("obj_mask")
    phys_key = [k for k in graphs[0] if k.endswith("_physics_param")]
    assert len(phys_key) == 1
    phys = {phys_key[0]: stack(phys_key[0]).reshape(B, -1)}
    n_valid = obj_mask.sum(1).clamp_min(1).float()
    errors = torch.zeros((B, T), device=dev)
    for t in range(T):
        # step 0 uses the start graph's own state/action; its edges are rebuilt here (same states -> same edges as graph['Rr'])
        edges = build_edges(state[:, -1], P["adj_thresh"], state_mask, eef_mask, P["topk"], P["connect_tool_all"], "single",
                            max_tools=n_eef)
        pred, _ = model(state, attrs, edges, None, p_instance, action=action, **phys)
        errors[:, t] = ((pred - gt[:, t]).norm(dim=-1) * obj_mask).sum(1) / n_valid
        if t + 1 < T:
            nxt = torch.cat([pred, eef_start[:, t + 1]], 1)
            state = torch.cat([state[:, 1:], nxt[:, None]], 1)
            action = torch.zeros_like(action)
            action[:, max_nobj:] = eef_delta[:, t + 1]
    errors = errors.cpu().numpy()
    return [[errors[b, t] for t in range(len(schedules[b]))] for b in range(B)]


def rollout_from_start_graph(graph, fps_idx_list, dataset_config, material_config, model, device, eef_pos, obj_pos,
                             current_start, current_end, get_next_pair_or_break_func, pairs, save_dir=None, viz=False,
                             imgs=None, cam_info=None):
    """Reference signature (rollout.py:20-143) for one start graph; `viz` output is not produced by this engine."""
    n_his = dataset_config["n_his"]
    assert eef_pos.shape[0] == obj_pos.shape[0]
    sched = frame_schedule(pairs, n_his, obj_pos.shape[0], current_start, current_end, get_next_pair_or_break_func)
    return rollout_batch(model, device, [graph], [fps_idx_list], [sched], [eef_pos], [obj_pos], dataset_config)[0]


def _episode_starts(dataset_config, material_config, eef_pos, obj_pos, episode_idx, pairs, physics_param, device):
    """Start graph + frame schedule of every push of an episode (the first row of each push file), rollout.py:149-165."""
    n_his = dataset_config["n_his"]
    pairs_path = os.path.join(dataset_config["prep_data_dir"], dataset_config["data_name"], "frame_pairs")
    out = []
    for path in sorted(glob.glob(os.path.join(pairs_path, f"{episode_idx:06}_*.txt"))):
        pair = np.loadtxt(path).astype(int)[0]
        eef_epi, obj_epi = eef_pos[episode_idx], obj_pos[episode_idx]
        graph, fps_idx_list = construct_graph(dataset_config, material_config, eef_epi, obj_epi, n_his, pair, physics_param,
                                              device=device)
        sched = frame_schedule(pairs, n_his, obj_epi.shape[0], pair[n_his - 1], pair[n_his], get_next_pair_or_break_episode_pushes)
        out.append((graph, fps_idx_list, sched, eef_epi, obj_epi))
    return out


def rollout_episode_pushes(model, device, dataset_config, material_config, eef_pos, obj_pos, episode_idx, pairs, physics_param,
                           save_dir, viz=False, imgs=None, cam_info=None):
    """-> [error list per push]; writes error_<i>.txt per push like the reference (rollout.py:145-196; no plots/videos)."""
    starts = _episode_starts(dataset_config, material_config, eef_pos, obj_pos, episode_idx, pairs, physics_param, device)
    errs = rollout_batch(model, device, *map(list, zip(*starts)), dataset_config) if starts else []
    for i, e in enumerate(errs):
        np.savetxt(os.path.join(save_dir, f"error_{i + 1}.txt"), np.array(e))
    return errs


def rollout_dataset(model, device, config, save_dir, viz=False):
    """Validation split -> per-episode/per-push error files + error_short.txt (steps x pushes, truncated to the shortest
    rollout), rollout.py:198-265.  All pushes of all episodes run as ONE batch.  -> step_error array."""
    dataset_config, material_config = config["dataset_config"], config["material_config"]
    pair_lists, physics_params = load_dataset(dataset_config, material_config, phase="valid")
    pair_lists = np.array(pair_lists)
    eef_pos, obj_pos = load_positions(dataset_config)
    starts, owners = [], []
    for episode_idx in sorted(np.unique(pair_lists[:, 0]).astype(int)):
        pairs_epi = pair_lists[pair_lists[:, 0] == episode_idx][:, 1:]
        epi = _episode_starts(dataset_config, material_config, eef_pos, obj_pos, episode_idx, pairs_epi, physics_params[episode_idx],
                              device)
        starts.extend(epi)
        owners.extend((episode_idx, i + 1) for i in range(len(epi)))
    total = rollout_batch(model, device, *map(list, zip(*starts)), dataset_config)
    for (episode_idx, push), e in zip(owners, total):
        d = os.path.join(save_dir, f"{episode_idx}", "short")
        os.makedirs(d, exist_ok=True)
        np.savetxt(os.path.join(d, f"error_{push}.txt"), np.array(e))
    min_step = min(len(e) for e in total)
    step_error = np.array([[e[i] for e in total] for i in range(min_step)], np.float64)
    np.savetxt(os.path.join(save_dir, "error_short.txt"), step_error)
    return step_error


def rollout(config, epoch, viz=False):
    """Entry point with the reference's config contract (rollout.py:267-305): loads
    <train_config.out_dir>/<data_name>/checkpoints/model_<epoch>.pth | latest.pth and evaluates the validation split."""
    from .model import DynamicsPredictor
    dataset_config, train_config = config["dataset_config"], config["train_config"]
    np.random.seed(train_config["random_seed"])
    torch.manual_seed(train_config["random_seed"])
    device = torch.device(dataset_config["device"])
    data_name = dataset_config["data_name"]
    save_dir = os.path.join(config["rollout_config"]["out_dir"], f"rollout-{data_name}-model_{epoch}")
    os.makedirs(save_dir, exist_ok=True)
    ckpt = "latest.pth" if epoch == "latest" else f"model_{epoch}.pth"
    model = DynamicsPredictor(config["model_config"], config["material_config"], dataset_config, device)
    model.to(device).eval()
    model.load_state_dict(torch.load(os.path.join(train_config["out_dir"], data_name, "checkpoints", ckpt), map_location=device))
    return rollout_dataset(model, device, config, save_dir, viz)
