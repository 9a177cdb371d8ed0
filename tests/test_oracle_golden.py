"""Pin the CPU oracle (oracle/) to the golden vectors captured from the imported reference.

SURVEY.md §8c: the reference has no tests of its own for this path, so these fixtures
(tools/gen_golden.py) are what "same results as the reference" means.
"""
import numpy as np
import pytest

from conftest import golden_files, load_golden, weights_for
from adaptigraph_amd import configs
from oracle import ag_oracle as ago

TOL = 2e-5   # fp32 summation-order noise between torch-CPU sgemm and the oracle's k-ordered fma chains


@pytest.mark.parametrize("name", golden_files("edges_"))
def test_edges_exact(name):
    g = load_golden(name)
    n_rel, recv, send = ago.build_edges(g["pos"], g["radius"], g["mask"], g["tool_mask"], int(g["topk"]),
                                        bool(g["connect_tools_all"]), str(g["variant"]), e_cap=g["recv"].shape[1] + 8)
    assert np.array_equal(n_rel, g["n_rel"])
    for b, n in enumerate(n_rel):
        assert np.array_equal(recv[b, :n], g["recv"][b, :n]), f"sample {b} receivers"
        assert np.array_equal(send[b, :n], g["send"][b, :n]), f"sample {b} senders"
        assert np.all(np.diff(recv[b, :n].astype(np.int64) * g["pos"].shape[1] + send[b, :n]) > 0)  # CSR order


@pytest.mark.parametrize("name", golden_files("fwd_"))
def test_forward(name, weights):
    g = load_golden(name)
    w = dict(weights_for(g, weights))
    s = float(g["decoder_scale"])
    if s != 1.0:
        w["non_rigid_predictor.linear_2.weight"] = w["non_rigid_predictor.linear_2.weight"] * np.float32(s)
        w["non_rigid_predictor.linear_2.bias"] = w["non_rigid_predictor.linear_2.bias"] * np.float32(s)
    pos, mot = ago.forward(w, g["state"], g["attrs"], g["action"], g["p_instance"], g["phys"], g["n_rel"], g["recv"],
                           g["send"])
    scale = max(1.0, float(np.abs(g["pred_motion"]).max()))
    assert np.abs(mot - g["pred_motion"]).max() <= TOL * scale
    assert np.abs(pos - g["pred_pos"]).max() <= TOL * scale
    if s != 1.0:   # the clamp (model.py:309) must have been exercised
        assert np.abs(g["pred_motion"]).max() > 100 and np.abs(pos - g["state"][:, -1, :pos.shape[1]]).max() <= 100.0


@pytest.mark.parametrize("name", ["fwd_rope301", "fwd_granular205", "fwd_cloth257"])
def test_dense_bmm_baseline_formulation_matches_reference(name, weights):
    """bench.py's second CPU baseline (oracle/torch_dense.py: one-hot Rr/Rs + bmm in PyTorch-CPU, the reference's formulation)
    reproduces the reference forward on the golden inputs, so what it times is the reference's algorithm."""
    import torch
    from adaptigraph_amd.model import _Dec, _Lin, _MLP3
    from oracle.torch_dense import dense_forward, one_hots
    g = load_golden(name)
    if float(g["decoder_scale"]) != 1.0:
        pytest.skip("plain weights only")
    m = torch.nn.Module()
    m.particle_encoder, m.relation_encoder = _MLP3(6, 150, 150), _MLP3(17, 150, 150)
    m.particle_propagator, m.relation_propagator, m.non_rigid_predictor = _Lin(300, 150), _Lin(450, 150), _Dec(150, 150, 3)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    Rr, Rs = one_hots(g["n_rel"], g["recv"], g["send"], g["attrs"].shape[1])
    with torch.no_grad():
        pos, mot = dense_forward(m, t(g["state"]), t(g["attrs"]), Rr, Rs, t(g["p_instance"]), t(g["action"]), t(g["phys"]))
    assert np.abs(mot.numpy() - g["pred_motion"]).max() <= TOL and np.abs(pos.numpy() - g["pred_pos"]).max() <= TOL


@pytest.mark.parametrize("name", golden_files("decode_action"))
def test_decode_action(name):
    g = load_golden(name)
    d, r = ago.decode_action(g["action"], float(g["push_length"]))
    assert np.array_equal(r, g["repeat"])
    assert np.abs(d - g["decoded"]).max() <= 1e-6


@pytest.mark.parametrize("name", golden_files("dyn_"))
def test_dynamics(name, weights):
    g = load_golden(name)
    task = configs.task_config(str(g["material"]))
    seq, dec = ago.dynamics(weights_for(g, weights), task, g["state"], g["action"])
    assert np.abs(dec - g["action_seqs"]).max() <= 1e-6
    assert np.abs(seq - g["state_seqs"]).max() <= 1e-4


@pytest.mark.parametrize("name", golden_files("dynmask_"))
def test_dynamics_masked(name, weights):
    g = load_golden(name)
    task = configs.task_config(str(g["material"]))
    seq, dec = ago.dynamics_masked(weights, task, g["state_init"], g["state_mask"], g["action"])
    assert np.abs(dec - g["action_seqs"]).max() <= 1e-6
    assert np.abs(seq - g["state_seqs"]).max() <= 1e-4


def test_top_k_tie_rule_is_lowest_index():
    # duplicate points => exactly equal distances; the oracle's documented rule keeps the lower sender index
    pos = np.zeros((1, 6, 3), np.float32)
    pos[0, 1:5, 0] = 0.1           # four senders at identical distance from particle 0
    pos[0, 5, 0] = 5.0
    mask = np.ones((1, 6), bool)
    tool = np.zeros((1, 6), bool)
    n, recv, send = ago.build_edges(pos, 0.5, mask, tool, topk=3, connect_tools_all=False, variant="batch")
    row0 = send[0, :n[0]][recv[0, :n[0]] == 0]
    assert row0.tolist() == [0, 1, 2]
