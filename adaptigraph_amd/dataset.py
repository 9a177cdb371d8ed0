"""Training / validation samples from the reference's preprocessed dataset layout — SURVEY.md §8f row n4 data side
(src/dynamics/dataset/dataset.py:10-252).

`DynDataset[idx]` yields the same tensors as the reference for the same numpy RNG state (FPS start indices, FPS radius,
physics noise, state noise, rotation, adjacency radius — drawn in that order), except that it does NOT build dense
`Rr`/`Rs` on the host: it returns the masks and the drawn radius, and `attach_edges` builds the whole batch's adjacency on
the GPU (`ag_build_edges`, single-graph rule variant, per-sample radius) after collation.
"""
import numpy as np
import torch
from torch.utils.data import Dataset

from .graph import build_edges
from .load import load_dataset, load_positions
from .sampling import fps
from .train_ops import EdgeViews


class DynDataset(Dataset):
    def __init__(self, dataset_config, material_config, phase="train"):
        assert phase in ["train", "valid"]
        self.phase = phase
        self.dataset_config, self.material_config = dataset_config, material_config
        self.verbose = dataset_config.get("verbose", False)
        self.n_his, self.n_future = dataset_config["n_his"], dataset_config["n_future"]
        rnd = dataset_config["randomness"]
        self.add_randomness = rnd["use"]
        self.state_noise, self.phys_noise = rnd["state_noise"][phase], rnd["phys_noise"][phase]
        assert len(dataset_config["datasets"]) == 1, "Only one object type is supported."
        d = self.dataset = dataset_config["datasets"][0]
        self.max_nobj, self.fps_radius_range = d["max_nobj"], d["fps_radius_range"]
        self.max_nR, self.adj_radius_range = d["max_nR"], d["adj_radius_range"]
        self.topk, self.connect_tool_all = d["topk"], d["connect_tool_all"]
        self.pair_lists, self.physics_params = load_dataset(dataset_config, material_config, phase)
        self.pair_lists = np.array(self.pair_lists)
        self.materials = {k: v.shape[0] for k, v in self.physics_params[0].items()}
        self.eef_pos, self.obj_pos = load_positions(dataset_config)
        self.pos_dim = self.obj_pos[0].shape[-1]
        self.obj_dim, self.eef_dim = self.max_nobj, self.eef_pos[0].shape[1]
        self.state_dim = self.obj_dim + self.eef_dim

    def __len__(self):
        return len(self.pair_lists)

    def __getitem__(self, idx):
        H, Fu, no, ns = self.n_his, self.n_future, self.obj_dim, self.state_dim
        epi = int(self.pair_lists[idx][0])
        pair = self.pair_lists[idx][1:].astype(int)
        assert len(pair) == H + Fu
        obj_kps = np.asarray(self.obj_pos[epi])[pair]            # (H+Fu, N_all, 3)
        eef_kps = np.asarray(self.eef_pos[epi])[pair]            # (H+Fu, N_eef, 3)
        fps_idx = fps(obj_kps[H - 1], self.max_nobj, self.fps_radius_range, verbose=self.verbose)
        n_kp, n_eef = len(fps_idx), eef_kps.shape[1]

        kp = np.zeros((H + Fu, no, self.pos_dim), np.float32)    # sampled key-points of every frame, zero-padded
        kp[:, :n_kp] = obj_kps[:, fps_idx]
        state_history = np.zeros((H, ns, self.pos_dim), np.float32)
        state_history[:, :no], state_history[:, no:] = kp[:H], eef_kps[:H]
        states_delta = np.zeros((ns, self.pos_dim), np.float32)
        states_delta[no:] = eef_kps[H] - eef_kps[H - 1]
        obj_kp_future = kp[H:].copy()
        eef_future = np.zeros((Fu - 1, ns, self.pos_dim), np.float32)
        states_delta_future = np.zeros((Fu - 1, ns, self.pos_dim), np.float32)
        eef_future[:, no:] = eef_kps[H:H + Fu - 1]
        states_delta_future[:, no:] = eef_kps[H + 1:H + Fu] - eef_kps[H:H + Fu - 1]

        state_mask = np.zeros(ns, bool)
        state_mask[:n_kp] = True
        state_mask[no:] = True
        eef_mask = np.zeros(ns, bool)
        eef_mask[no:] = True
        attrs = np.zeros((ns, 2), np.float32)
        attrs[:n_kp, 0] = 1.0
        attrs[no:, 1] = 1.0
        p_instance = np.zeros((no, 1), np.float32)
        p_instance[:n_kp, 0] = 1

        physics_param = self.physics_params[epi]                 # noise accumulates in place, as in the reference (:175-179)
        for m in self.dataset_config["materials"]:
            if m not in physics_param.keys():
                raise ValueError(f"Physics parameter {m} not found in {self.dataset_config['data_dir']}")
            physics_param[m] += np.random.uniform(-self.phys_noise, self.phys_noise, size=physics_param[m].shape)
        assert len(self.dataset_config["materials"]) == 1, "only support single material"
        material_idx = np.zeros((no, len(self.material_config["material_index"])), np.int64)
        material_idx[:n_kp, self.material_config["material_index"][self.dataset_config["materials"][0]]] = 1

        if self.add_randomness:                                   # position noise, then one rotation about the last axis
            state_history += np.random.uniform(-self.state_noise, self.state_noise, size=state_history.shape)   # stays fp32
            ang = np.random.uniform(-np.pi, np.pi)
            rot = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]], dtype=state_history.dtype)
            state_history, states_delta = state_history @ rot[None], states_delta @ rot
            eef_future, states_delta_future, obj_kp_future = eef_future @ rot[None], states_delta_future @ rot[None], obj_kp_future @ rot[None]
        adj_thresh = np.random.uniform(*self.adj_radius_range)

        f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()
        graph = {"state": f32(state_history), "action": f32(states_delta), "eef_future": f32(eef_future),
                 "action_future": f32(states_delta_future), "state_future": f32(obj_kp_future), "attrs": f32(attrs),
                 "p_rigid": torch.zeros(1), "p_instance": f32(p_instance), "obj_mask": torch.from_numpy(np.arange(no) < n_kp),
                 "state_mask": torch.from_numpy(state_mask), "eef_mask": torch.from_numpy(eef_mask),
                 "material_index": torch.from_numpy(material_idx), "adj_thresh": torch.tensor(adj_thresh, dtype=torch.float64)}
        for m, dim in self.materials.items():
            graph[m + "_physics_param"] = f32(physics_param[m]) if m in physics_param else torch.zeros(dim)
        return graph


def attach_edges(data, dataset_config, device):
    """Collated batch (CPU or GPU tensors) -> the same dict on `device` with the adjacency of every sample built there:
    data['Rr'] = CSREdges (and data['Rs'] = None, data['edge_views'] for the training ops)."""
    d = dataset_config["datasets"][0]
    data = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in data.items()}
    radii = data.pop("adj_thresh").double().cpu().numpy().reshape(-1)
    csr = build_edges(data["state"][:, -1].contiguous(), radii, data["state_mask"], data["eef_mask"], d["topk"], d["connect_tool_all"],
                      "single", max_tools=int(data["eef_mask"].shape[1] - d["max_nobj"]))
    data["Rr"], data["Rs"], data["edge_views"] = csr, None, EdgeViews(csr)
    return data
