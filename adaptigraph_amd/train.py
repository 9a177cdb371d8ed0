"""Training loop on MI355X — SURVEY.md §8f row n4 (src/dynamics/train/train.py:19-148, without the loss plot).

Same config contract (dataset_config / train_config / model_config / material_config), optimiser (Adam, lr 1e-3), objective
(`unrolled_loss`: n_future-step MSE with the prediction fed back), phase schedule and checkpoint files
(`<out_dir>/<data_name>/checkpoints/{model_<epoch>.pth, latest.pth, latest_optim.pth}`).  Differences by design: the batch's
adjacency is built on the GPU after collation (`attach_edges`) instead of densely per sample in the loader workers, and the
model is `TrainableDynamicsPredictor` (CSR gather / segment-reduce HIP kernels + library GEMMs).
"""
import os
import random

import numpy as np
import torch
from torch.utils.data import DataLoader

from . import train_ops

from .dataset import DynDataset, attach_edges
from .train_model import TrainableDynamicsPredictor, unrolled_loss


def set_seed(seed):
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


def _cycle(loader):
    while True:
        for batch in loader:
            yield batch


def train(config):
    """-> {'train': [mean logged loss per epoch], 'valid': [...]}."""
    dataset_config, train_config = config["dataset_config"], config["train_config"]
    model_config, material_config = config["model_config"], config["material_config"]
    data_name = dataset_config["data_name"]
    ckpt_dir = os.path.join(train_config["out_dir"], data_name, "checkpoints")
    os.makedirs(ckpt_dir, exist_ok=True)
    set_seed(train_config["random_seed"])
    if not torch.cuda.is_available():
        raise RuntimeError("adaptigraph_amd.train needs an MI355X: the graph kernels have no CPU path")
    device = torch.device(dataset_config.get("device", "cuda") if str(dataset_config.get("device", "cuda")).startswith("cuda") else "cuda")

    n_future, phases = dataset_config["n_future"], train_config["phases"]
    datasets = {ph: DynDataset(dataset_config, material_config, phase=ph) for ph in phases}
    loaders = {ph: _cycle(DataLoader(datasets[ph], batch_size=train_config["batch_size"], shuffle=(ph == "train"),
                                     num_workers=train_config["num_workers"])) for ph in phases}
    model = TrainableDynamicsPredictor(model_config, material_config, dataset_config, device).to(device)
    optimizer = torch.optim.Adam(model.parameters(), lr=0.001)

    history = {ph: [] for ph in phases}
    for epoch in range(train_config["n_epochs"]):
        for ph in phases:
            model.train(ph == "train")
            n_iters = train_config["n_iters_per_epoch"][ph]
            n_iters = len(datasets[ph]) if n_iters == -1 else n_iters
            logged = []
            with torch.set_grad_enabled(ph == "train"):
                for i in range(n_iters):
                    data = attach_edges(next(loaders[ph]), dataset_config, device)
                    if ph == "train":
                        optimizer.zero_grad(set_to_none=False)      # the gradient kernel accumulates into the kept .grad buffers
                    # parameter gradients straight into .grad for this forward + backward only (train_ops.DIRECT_GRADS is a
                    # process-wide switch: code that asks autograd for parameter gradients must find it off)
                    with train_ops.direct_grads(ph == "train" and getattr(model, "fused_dense", False)):
                        loss = unrolled_loss(model, data, n_future)
                        if ph == "train":
                            loss.backward()
                    if ph == "train":
                        optimizer.step()
                        if i % train_config["log_interval"] == 0:
                            logged.append(loss.item())
                            print(f"Epoch {epoch}, iter {i}, loss {logged[-1]}")
                    else:
                        logged.append(loss.item())
            history[ph].append(float(np.mean(logged)) if logged else float("nan"))
            if ph == "valid":
                print(f"\nEpoch {epoch}, valid loss {history[ph][-1]}")
        done = epoch + 1
        if (done < 100 and done % 10 == 0) or done % 100 == 0:
            torch.save(model.state_dict(), os.path.join(ckpt_dir, f"model_{done}.pth"))
        torch.save(model.state_dict(), os.path.join(ckpt_dir, "latest.pth"))
        torch.save(optimizer.state_dict(), os.path.join(ckpt_dir, "latest_optim.pth"))
    return history
