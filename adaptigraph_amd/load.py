"""Readers for the reference's preprocessed dataset layout — SURVEY.md §8f row n3 (src/dynamics/dataset/load.py:6-83).

    <prep_data_dir>/<data_name>/frame_pairs/<episode:06>_<push:02>.txt   rows of n_his + n_future frame indices
    <prep_data_dir>/<data_name>/positions.pkl                            {'eef_pos': [ (T,N_eef,3) ], 'obj_pos': [ (T,N_obj,3) ]}
    <data_dir>/<data_name>/<episode:06>/property_params.pkl              {param name: raw value}
"""
import glob
import os
import pickle

import numpy as np


def load_pairs(pairs_path, episode_range):
    """-> int array (n_pairs, 1 + n_his + n_future): episode index, then the frame indices.  Push files holding a single
    row load as 1-D and are skipped, as in the reference (:12)."""
    rows = []
    for epi in episode_range:
        n_pushes = len(glob.glob(os.path.join(pairs_path, f"{epi:06}_*.txt")))
        for push in range(1, n_pushes + 1):
            frames = np.loadtxt(os.path.join(pairs_path, f"{epi:06}_{push:02}.txt"))
            if frames.ndim == 1:
                continue
            rows.extend(np.concatenate([np.full((frames.shape[0], 1), epi, frames.dtype), frames], axis=1))
    return np.array(rows).astype(int)


def normalized_physics_params(properties, dataset_config, material_config):
    """Raw episode properties -> {material: fp32 vector of the used params scaled to [0,1] by their config range} (:46-62)."""
    out = {}
    for material in dataset_config["materials"]:
        vals = [(properties[p["name"]] - p["min"]) / (p["max"] - p["min"] + 1e-6)
                for p in material_config[material]["physics_params"] if p["name"] in properties.keys() and p["use"]]
        out[material] = np.array(vals).astype(np.float32)
    return out


def load_dataset(dataset_config, material_config, phase="train"):
    """-> (pair_lists of the phase's episode slice, physics params of EVERY episode)."""
    name = dataset_config["data_name"]
    data_dir = os.path.join(dataset_config["data_dir"], name)
    prep_dir = os.path.join(dataset_config["prep_data_dir"], name)
    num_epis = sum(1 for f in os.listdir(data_dir) if f.isdigit() and os.path.isdir(os.path.join(data_dir, f)))
    lo, hi = dataset_config["ratio"][phase]
    pair_lists = load_pairs(os.path.join(prep_dir, "frame_pairs"), range(int(num_epis * lo), int(num_epis * hi)))
    physics_params = []
    for epi in range(num_epis):
        with open(os.path.join(data_dir, f"{epi:06}/property_params.pkl"), "rb") as f:
            physics_params.append(normalized_physics_params(pickle.load(f), dataset_config, material_config))
    return pair_lists, physics_params


def load_positions(dataset_config):
    with open(os.path.join(dataset_config["prep_data_dir"], dataset_config["data_name"], "positions.pkl"), "rb") as f:
        positions = pickle.load(f)
    return positions["eef_pos"], positions["obj_pos"]
