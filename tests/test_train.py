"""Training path ("next" row n4): dataset samples, differentiable graph ops, the n_future unroll and its gradients, the
training loop — against fixtures generated from the reference (tests/golden/train_rope.npz)."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from adaptigraph_amd import configs
from oracle import ag_oracle as ago
from test_eval_rollout import make_config, write_dataset

KEYS = ["state", "action", "eef_future", "action_future", "state_future", "attrs", "p_instance", "obj_mask", "rope_physics_param"]


def train_config(root, g_eval, device="cpu"):
    cfg = make_config(root, g_eval, device)
    ds = cfg["dataset_config"]
    ds.update(ratio={"train": [0, 0.67], "valid": [0.67, 1.0]}, verbose=False,
              randomness={"use": True, "state_noise": {"train": 0.05, "valid": 0.0}, "phys_noise": {"train": 0.0, "valid": 0.0}})
    cfg["train_config"].update(phases=["train", "valid"], num_workers=0, batch_size=4, n_epochs=2, log_interval=1,
                               n_iters_per_epoch={"train": 3, "valid": 2})
    return cfg


@pytest.fixture()
def dataset_dir(tmp_path):
    g_eval = load_golden("evalrollout_rope")
    write_dataset(str(tmp_path), g_eval)
    return g_eval, str(tmp_path)


def batch_from(g):
    return {k: g["b_" + k] for k in KEYS}


def test_dataset_samples_match_reference(dataset_dir):
    from adaptigraph_amd.dataset import DynDataset
    g_eval, root = dataset_dir
    g = load_golden("train_rope")
    cfg = train_config(root, g_eval)
    dset = DynDataset(cfg["dataset_config"], cfg["material_config"], phase="train")
    assert len(dset) == int(g["n_samples"])
    for k, i in enumerate(g["idx"]):
        np.random.seed(int(g["seed"]) + k)
        s = dset[int(i)]
        for key in KEYS:
            assert np.array_equal(s[key].numpy(), g["b_" + key][k]), (key, k)
        assert s["state_mask"].sum() == s["obj_mask"].sum() + 1 and s["eef_mask"].sum() == 1
        assert 0.48 <= float(s["adj_thresh"]) <= 0.52
    valid = DynDataset(cfg["dataset_config"], cfg["material_config"], phase="valid")
    assert len(valid) == 72 - len(dset)


def test_oracle_unrolled_loss_matches_reference(weights):
    g = load_golden("train_rope")
    loss, preds = ago.unrolled_loss(weights, batch_from(g), g["n_rel"], g["recv"], g["send"])
    assert abs(loss - float(g["loss"])) <= 1e-6 and np.abs(preds - g["preds"]).max() <= 2e-5


def test_train_ops_refuse_cpu():
    from adaptigraph_amd import train_ops
    with pytest.raises(RuntimeError, match="no CPU path"):
        train_ops._GatherRows.apply(torch.zeros(3, 4), torch.zeros(2, dtype=torch.int32), torch.zeros(4, dtype=torch.int32), None)


# ------------------------------------------------------------------------------------------------------ GPU
DEV = "cuda:0"


def tg(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def golden_csr(g, N):
    from test_gpu_parity import csr_from_lists
    return csr_from_lists(g["n_rel"], g["recv"], g["send"], N)


def trainable(weights):
    from adaptigraph_amd.train_model import TrainableDynamicsPredictor
    m = TrainableDynamicsPredictor(configs.model_config(), configs.material_config("rope"), configs.dataset_config("rope"), DEV)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})
    return m.to(DEV)


@pytest.mark.gpu
def test_graph_ops_forward_and_adjoint_vs_torch():
    """gather / message_sum and their hand-written backward kernels against plain torch indexing + autograd (fp64 reference)."""
    from adaptigraph_amd import graph as aggraph, train_ops
    rng = np.random.default_rng(0)
    B, N, D = 3, 37, 150
    pos = rng.uniform(0, 2.0, (B, N, 3)).astype(np.float32)
    mask = np.ones((B, N), bool); mask[1, 30:] = False
    tool = np.zeros((B, N), bool); tool[:, -1] = True
    csr = aggraph.build_edges(tg(pos), 0.6, tg(mask), tg(tool), 6, False, "batch", max_tools=1)
    v = train_ops.EdgeViews(csr)
    assert v.E > 50 and int(v.col_ptr[-1]) == v.E
    recv, send = v.recv.long(), v.send.long()
    M = B * N
    x = torch.randn(M, D, device=DEV, requires_grad=True)
    e = torch.randn(v.E, D, device=DEV, requires_grad=True)
    hr = torch.randn(M, D, device=DEV, requires_grad=True)
    w1, w2 = torch.randn(v.E, D, device=DEV), torch.randn(M, D, device=DEV)
    out = (train_ops.gather_receivers(x, v) * w1).sum() + (train_ops.gather_senders(x, v) * w1.flip(0)).sum() \
        + (train_ops.message_sum(e, hr, x, v) * w2).sum()
    gx, ge, ghr = torch.autograd.grad(out, (x, e, hr))
    xd, ed, hrd = (t.detach().double().requires_grad_() for t in (x, e, hr))
    msg = torch.relu(ed + hrd[recv] + xd[send])
    ref = (xd[recv] * w1.double()).sum() + (xd[send] * w1.flip(0).double()).sum() \
        + (torch.zeros(M, D, dtype=torch.float64, device=DEV).index_add(0, recv, msg) * w2.double()).sum()
    rx, re, rhr = torch.autograd.grad(ref, (xd, ed, hrd))
    assert abs(out.item() - ref.item()) <= 1e-4 * abs(ref.item())
    for a, b in ((gx, rx), (ge, re), (ghr, rhr)):
        assert (a.double() - b).abs().max().item() <= 1e-5 * b.abs().max().item()
    g2 = torch.autograd.grad((train_ops.message_sum(e, hr, x, v) * w2).sum(), (x, e, hr))      # fixed-order reductions: bitwise repeatable
    g3 = torch.autograd.grad((train_ops.message_sum(e, hr, x, v) * w2).sum(), (x, e, hr))
    assert all(torch.equal(a, b) for a, b in zip(g2, g3))


@pytest.mark.gpu
@pytest.mark.parametrize("precision", [1, 0])
@pytest.mark.parametrize("kind,d_in,rows", [("edge", 17, 1000), ("node", 6, 300), ("decoder", 150, 257), ("edge", 17, 1)])
def test_fused_dense_chain_forward_and_backward_vs_torch(kind, d_in, rows, precision, monkeypatch):
    """The fused MFMA chain kernels (forward saving every layer, backward with the ReLU masks in registers, weight gradients on
    the split-K MFMA kernel), in both arithmetic modes (split-bf16 default, exact fp32), against the same stack in plain torch
    (fp64 autograd): output, input gradient and every weight / bias gradient; a row count that leaves a partial 128-row tile;
    repeated calls are bitwise identical."""
    import torch.nn.functional as F
    from adaptigraph_amd import train_ops
    monkeypatch.setattr(train_ops, "CHAIN_PRECISION", precision)
    tol = 2e-5 if precision == 0 else 1e-4          # split-bf16: 2^-17 relative operand error (gradient gate of the golden: 2e-4)
    torch.manual_seed(3)
    n_layers = train_ops.CHAIN_KINDS[kind][1]
    dims = [d_in] + [150] * (n_layers - 1) + [3 if kind == "decoder" else 150]
    relu = [True] * (n_layers - 1) + [kind == "node"]
    big = torch.randn(150, 450, device=DEV) / 12          # the edge chain's last layer is a column slice of a wider parameter
    Ws = [(torch.randn(dims[l + 1], dims[l], device=DEV) / np.sqrt(dims[l])).requires_grad_() for l in range(n_layers)]
    if kind == "edge":
        big.requires_grad_()
        Ws[-1] = big
    bs = [(torch.randn(dims[l + 1], device=DEV) * 0.1).requires_grad_() for l in range(n_layers)]
    x = torch.randn(rows, d_in, device=DEV, requires_grad=True)
    probe = torch.randn(rows, dims[-1], device=DEV)

    def layers():
        return [((big[:, :150] if (kind == "edge" and l == n_layers - 1) else Ws[l]), bs[l]) for l in range(n_layers)]

    y = train_ops.fused_chain(kind, x, layers())
    got = torch.autograd.grad((y * probe).sum(), [x] + Ws + bs)
    xd = x.detach().double().requires_grad_()
    Wd = [w.detach().double().requires_grad_() for w in Ws]
    bd = [b.detach().double().requires_grad_() for b in bs]
    h = xd
    for l in range(n_layers):
        w = Wd[l][:, :150] if (kind == "edge" and l == n_layers - 1) else Wd[l]
        h = F.linear(h, w, bd[l])
        h = torch.relu(h) if relu[l] else h
    ref = torch.autograd.grad((h * probe.double()).sum(), [xd] + Wd + bd)
    assert y.shape == h.shape and (y.double() - h).abs().max().item() <= tol * max(1.0, h.abs().max().item())
    for a, b in zip(got, ref):
        assert a.shape == b.shape and (a.double() - b).abs().max().item() <= tol * max(1e-3, b.abs().max().item())
    y2 = train_ops.fused_chain(kind, x, layers())
    got2 = torch.autograd.grad((y2 * probe).sum(), [x] + Ws + bs)
    assert torch.equal(y, y2) and all(torch.equal(a, b) for a, b in zip(got, got2))
    with torch.no_grad():                                    # an in-place parameter update (optimiser step) invalidates the packed streams
        Ws[0].mul_(0.5)
    y3 = train_ops.fused_chain(kind, x, layers())
    assert not torch.equal(y, y3)


@pytest.mark.gpu
def test_pack_cache_is_keyed_on_tensor_identity_not_only_address():
    """A model that is freed and re-created lands in the same allocator blocks with the same version counters: the
    cached weight pack of the old model must not be served to the new one; `.data` writes need invalidate_packs()."""
    import gc
    from adaptigraph_amd import train_ops
    torch.manual_seed(3)
    x = torch.randn(200, 150, device=DEV)

    def run(scale):
        Ws = [(torch.randn(150, 150, device=DEV) * 0.1 * scale).requires_grad_() for _ in range(2)] + [(torch.randn(3, 150, device=DEV) * scale).requires_grad_()]
        bs = [torch.zeros(150, device=DEV).requires_grad_() for _ in range(2)] + [torch.zeros(3, device=DEV).requires_grad_()]
        y = train_ops.fused_chain("decoder", x, [(w, b) for w, b in zip(Ws, bs)]).detach().clone()
        return y, [w.data_ptr() for w in Ws], Ws, bs

    y1, ptr1, Ws, bs = run(1.0)
    del Ws, bs
    gc.collect()
    y2, ptr2, Ws, bs = run(2.0)                      # usually the very same blocks (caching allocator), version 0 again
    assert not torch.equal(y1, y2), f"stale pack served (same addresses: {ptr1 == ptr2})"
    Ws[0].data.mul_(0.0)                             # invisible to the version counter
    train_ops.invalidate_packs()
    y3 = train_ops.fused_chain("decoder", x, [(w, b) for w, b in zip(Ws, bs)]).detach()
    assert not torch.equal(y2, y3)


@pytest.mark.gpu
def test_unrolled_loss_and_gradients_match_reference(weights):
    from adaptigraph_amd.train_model import unrolled_loss
    g = load_golden("train_rope")
    model = trainable(weights).train()
    data = {k: tg(v) for k, v in batch_from(g).items()}
    data.update(Rr=golden_csr(g, g["b_attrs"].shape[1]), Rs=None)
    before = {k: v.clone() for k, v in data.items() if torch.is_tensor(v)}
    loss = unrolled_loss(model, data, 3)
    assert abs(loss.item() - float(g["loss"])) <= 2e-6 * max(1.0, float(g["loss"]))
    assert all(torch.equal(data[k], v) for k, v in before.items())          # the batch dict is left untouched
    loss.backward()
    for name, p in model.named_parameters():
        ref = g["grad_" + name]
        assert np.abs(p.grad.cpu().numpy() - ref).max() <= 2e-4 * np.abs(ref).max() + 1e-9, name
    torch.optim.Adam(model.parameters(), lr=0.001).step()
    key = "particle_encoder.model.0.weight"
    assert np.abs(model.state_dict()[key].cpu().numpy() - g["adam_" + key]).max() <= 1e-5


@pytest.mark.gpu
def test_edge_inputs_op_forward_and_adjoint_vs_torch():
    """train_ops.edge_inputs (rel_inputs of model.py:220-253 in one kernel, adjoint in two) against the torch composition it
    replaces (gathers + cat + abs + sum), values bit-exact and gradients vs fp64 autograd; two instance columns, repeatable."""
    from adaptigraph_amd import graph as aggraph, train_ops
    rng = np.random.default_rng(1)
    B, N, A, G, S = 3, 41, 2, 2, 12
    pos = rng.uniform(0, 2.0, (B, N, 3)).astype(np.float32)
    mask = np.ones((B, N), bool); mask[2, 35:] = False
    tool = np.zeros((B, N), bool); tool[:, -1] = True
    v = train_ops.EdgeViews(aggraph.build_edges(tg(pos), 0.6, tg(mask), tg(tool), 6, False, "batch", max_tools=1))
    tab = torch.randn(B * N, A + G + S, device=DEV)
    tab[:, A:A + G] = (torch.rand(B * N, G, device=DEV) > 0.5).float()          # group columns are 0/1 in the model: |diff| has kinks at 0
    tab.requires_grad_()
    probe = torch.randn(v.E, 2 * A + 1 + S, device=DEV)
    out = train_ops.edge_inputs(tab, v, A, G)
    (g1,) = torch.autograd.grad((out * probe).sum(), tab)
    td = tab.detach().double().requires_grad_()
    r, s_ = v.recv.long(), v.send.long()
    ref = torch.cat([td[r, :A], td[s_, :A], (td[r, A:A + G] - td[s_, A:A + G]).abs().sum(1, keepdim=True), td[r, A + G:] - td[s_, A + G:]], 1)
    (g2,) = torch.autograd.grad((ref * probe.double()).sum(), td)
    assert torch.equal(out, ref.float()) and (g1.double() - g2).abs().max().item() <= 1e-5 * g2.abs().max().item()
    (g3,) = torch.autograd.grad((train_ops.edge_inputs(tab, v, A, G) * probe).sum(), tab)
    assert torch.equal(g1, g3)


@pytest.mark.gpu
def test_direct_gradient_accumulation_equals_autograd(weights, monkeypatch):
    """train_ops.DIRECT_GRADS: the weight-gradient kernel accumulates straight into the leaf parameters' .grad (through the
    column slices of relation_propagator / particle_propagator too) and autograd sees None — the result must equal what
    autograd's own accumulation produces, over two backward passes (accumulation) of the 3-step unrolled loss."""
    from adaptigraph_amd import train_ops
    from adaptigraph_amd.train_model import unrolled_loss
    g = load_golden("train_rope")
    data = {k: tg(v) for k, v in batch_from(g).items()}
    data.update(Rr=golden_csr(g, g["b_attrs"].shape[1]), Rs=None)
    grads = {}
    for direct in (False, True):
        monkeypatch.setattr(train_ops, "DIRECT_GRADS", direct)
        model = trainable(weights).train()
        for _ in range(2):
            unrolled_loss(model, data, 3).backward()
        grads[direct] = {n: p.grad.clone() for n, p in model.named_parameters()}
    for n, ref in grads[False].items():
        assert (grads[True][n] - ref).abs().max().item() <= 1e-6 * max(1e-6, ref.abs().max().item()), n
    ref1 = g["grad_particle_encoder.model.0.weight"]
    assert np.abs(grads[True]["particle_encoder.model.0.weight"].cpu().numpy() - 2 * ref1).max() <= 4e-4 * np.abs(ref1).max()


@pytest.mark.gpu
def test_trainable_forward_equals_fused_engine(weights):
    """Same weights, same graph: the autograd forward and the fused inference engine (exact-fp32 mode) agree."""
    from adaptigraph_amd.model import DynamicsPredictor
    g = load_golden("train_rope")
    data = {k: tg(v) for k, v in batch_from(g).items()}
    csr = golden_csr(g, g["b_attrs"].shape[1])
    args = (data["state"], data["attrs"], csr, None, data["p_instance"])
    kw = dict(action=data["action"], rope_physics_param=data["rope_physics_param"])
    with torch.no_grad():
        pos_t, mot_t = trainable(weights).eval()(*args, **kw)
    eng = DynamicsPredictor(configs.model_config(), configs.material_config("rope"), configs.dataset_config("rope"), DEV)
    eng.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})
    pos_e, mot_e = eng.to(DEV).eval().set_option("precision", 0)(*args, **kw)
    assert (mot_t - mot_e).abs().max().item() <= 2e-5 and (pos_t - pos_e).abs().max().item() <= 2e-5
    assert np.abs(pos_t.cpu().numpy() - g["preds"][0]).max() <= 2e-5


@pytest.mark.gpu
def test_attach_edges_matches_reference_edges(dataset_dir):
    from adaptigraph_amd.dataset import DynDataset, attach_edges
    g_eval, root = dataset_dir
    g = load_golden("train_rope")
    cfg = train_config(root, g_eval, DEV)
    dset = DynDataset(cfg["dataset_config"], cfg["material_config"], phase="train")
    samples = []
    for k, i in enumerate(g["idx"]):
        np.random.seed(int(g["seed"]) + k)
        samples.append(dset[int(i)])
    data = attach_edges(torch.utils.data.default_collate(samples), cfg["dataset_config"], DEV)
    assert data["Rr"].n_rel().cpu().tolist() == g["n_rel"].tolist()
    for b, (r, s) in enumerate(data["Rr"].to_lists()):
        assert np.array_equal(r, g["recv"][b, :g["n_rel"][b]]) and np.array_equal(s, g["send"][b, :g["n_rel"][b]])


@pytest.mark.gpu
def test_training_loop_learns_and_checkpoints_load_into_engine(dataset_dir):
    from adaptigraph_amd import train as agtrain
    from adaptigraph_amd.model import DynamicsPredictor
    g_eval, root = dataset_dir
    cfg = train_config(root, g_eval, DEV)
    cfg["train_config"].update(n_epochs=10, n_iters_per_epoch={"train": 6, "valid": 2}, batch_size=8, random_seed=1)
    hist = agtrain.train(cfg)
    assert len(hist["train"]) == 10 and len(hist["valid"]) == 10 and np.isfinite(hist["train"]).all()
    assert np.mean(hist["train"][-3:]) < 0.7 * hist["train"][0]            # the objective goes down
    ck = os.path.join(cfg["train_config"]["out_dir"], "rope", "checkpoints")
    assert sorted(os.listdir(ck)) == ["latest.pth", "latest_optim.pth", "model_10.pth"]
    sd = torch.load(os.path.join(ck, "model_10.pth"), map_location="cpu")
    assert len(sd) == 22
    eng = DynamicsPredictor(cfg["model_config"], cfg["material_config"], cfg["dataset_config"], DEV)
    eng.load_state_dict(sd)                                                   # trained weights drop into the fused engine
