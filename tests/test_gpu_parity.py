"""Parity tests proper: the HIP path (through the C ABI) against the golden vectors captured from the
reference and against the CPU oracle on the same seeded inputs.  Needs an MI355X (`-m gpu`).

Bars: edge lists bit-exact; fp32 outputs within 1e-4 max-abs of the reference forward on identical graphs
(BASELINE.json north_star); measured deviation is ~1e-6, the tolerances below keep a margin but would catch
any structural error (a wrong edge or column block moves outputs by >= 1e-2).
"""
import os
import warnings

import numpy as np
import pytest
import torch

from conftest import golden_files, load_golden, weights_for
from adaptigraph_amd import _lib, configs, synth
from adaptigraph_amd import graph as aggraph
from adaptigraph_amd.forward_dynamics import dynamics, dynamics_masked, rollout
from adaptigraph_amd.model import DynamicsPredictor
from oracle import ag_oracle as ago

pytestmark = pytest.mark.gpu
TOL_FWD = 1e-4        # north_star gate (single forward, identical graphs)
TOL_TIGHT = 2e-5      # 5x inside the gate: exact-fp32 mode measures ~1e-7..1e-6, split-bf16 mode ~1e-6..6e-6
TOL_BY_PREC = {"f32": 2e-5, "bf16x3": 2e-5, "fast": 4e-5}   # gate 1e-4.  "fast" (q16 Eterm table + fp16 edge stack with residual bytes) measures <= 1.87e-5
                                                             # (scaled-decoder clamp golden; the others <= 6e-6, profiles/r04_fwd_err.txt): 2x margin over
                                                             # the measured worst case, so a toolchain scheduling change cannot flip a test 2.5x inside the gate
DEV = "cuda:0"


def t(x, dtype=None):
    r = torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
    return r if dtype is None else r.to(dtype)


PRECISIONS = {"fast": 2, "bf16x3": 1, "f32": 0}   # engine arithmetic modes (include/adaptigraph_hip.h "precision")


@pytest.fixture(scope="module", params=list(PRECISIONS))
def prec(request):
    return request.param


@pytest.fixture(scope="module")
def model(weights, prec):
    return make_model(weights, prec=prec)


def make_model(weights, material="rope", decoder_scale=1.0, prec="fast"):
    m = DynamicsPredictor(configs.model_config(), configs.material_config(material), configs.dataset_config(material), DEV)
    sd = {k: torch.from_numpy(v.copy()) for k, v in weights.items()}
    if decoder_scale != 1.0:
        sd["non_rigid_predictor.linear_2.weight"] *= decoder_scale
        sd["non_rigid_predictor.linear_2.bias"] *= decoder_scale
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    m.set_option("precision", PRECISIONS[prec])
    return m


def lists_from(n_rel, recv, send):
    return [(recv[b, :n], send[b, :n]) for b, n in enumerate(n_rel)]


def csr_from_lists(n_rel, recv, send, N):
    """golden/oracle per-sample edge lists -> CSREdges on the GPU (edges are already receiver-sorted)."""
    B = len(n_rel)
    r = np.concatenate([recv[b, :n] + b * N for b, n in enumerate(n_rel)]).astype(np.int32)
    s = np.concatenate([send[b, :n] + b * N for b, n in enumerate(n_rel)]).astype(np.int32)
    row_ptr = np.zeros(B * N + 1, np.int32)
    np.add.at(row_ptr, r + 1, 1)
    row_ptr = np.cumsum(row_ptr).astype(np.int32)
    e = max(len(r), 1)
    rr = np.zeros(e, np.int32)
    ss = np.zeros(e, np.int32)
    rr[:len(r)] = r
    ss[:len(s)] = s
    return aggraph.CSREdges(t(row_ptr), t(rr), t(ss), B, N, len(r))


# ------------------------------------------------------------------------------------------ edge builder
@pytest.mark.parametrize("name", golden_files("edges_"))
def test_edges_golden_exact(name):
    g = load_golden(name)
    radius = t(g["radius"].astype(np.float32)) if bool(g["radius_is_tensor"]) else float(g["radius"][0])
    csr = aggraph.build_edges(t(g["pos"]), radius, t(g["mask"]), t(g["tool_mask"]), int(g["topk"]),
                              bool(g["connect_tools_all"]), str(g["variant"]), max_tools=int(g["tool_mask"].sum(1).max()))
    got = csr.to_lists()
    assert csr.n_rel().cpu().tolist() == g["n_rel"].tolist()
    for b, (r, s) in enumerate(got):
        n = g["n_rel"][b]
        assert np.array_equal(r, g["recv"][b, :n]) and np.array_equal(s, g["send"][b, :n]), f"sample {b}"


@pytest.mark.parametrize("material,n_obj,batch,variant,kw", [
    ("rope", 1000, 3, "batch", dict(spacing=0.1)),            # BASELINE configs[1] graph size
    ("rope", 1000, 2, "single", dict(spacing=0.1)),
    ("granular", 2000, 2, "batch", {}),                        # configs[2]: top-k 20 saturated, 5 tools
    ("cloth", 4096, 2, "batch", {}),                           # configs[3]: connect_tools_all, N = 4097
    ("cloth", 4096, 1, "single", {}),
    ("cloth", 1024, 2, "batch", dict(tool_near=False)),
    ("rope", 700, 2, "batch", dict(dense=True)),               # dense blob: > 384 in-radius candidates per row (prune path)
])
def test_edges_vs_oracle_exact(material, n_obj, batch, variant, kw):
    dense = kw.pop("dense", False) if isinstance(kw, dict) else False
    kw = {k: v for k, v in kw.items() if k != "dense"}
    g = synth.make_graph_inputs(material, n_obj, batch, seed=21, **kw)
    m = synth.MATERIALS[material]
    pos = g["state"][:, -1].copy()
    if dense:
        pos[:, :n_obj] = np.random.default_rng(5).uniform(0, 0.4, (batch, n_obj, 3)).astype(np.float32)
    n_rel, recv, send = ago.build_edges(pos, m["radius"], g["mask"], g["tool_mask"], m["topk"], m["connect_tools_all"], variant)
    csr = aggraph.build_edges(t(pos), m["radius"], t(g["mask"]), t(g["tool_mask"]), m["topk"], m["connect_tools_all"],
                              variant, max_tools=g["n_tools"])
    assert csr.n_rel().cpu().tolist() == n_rel.tolist()
    for b, (r, s) in enumerate(csr.to_lists()):
        assert np.array_equal(r, recv[b, :n_rel[b]]) and np.array_equal(s, send[b, :n_rel[b]]), f"sample {b}"


@pytest.mark.parametrize("material,n_obj,topk,variant", [
    ("rope", 300, 10, "batch"),          # connect_tools_all with top-k > 8: finalize_connect_kernel's generic merge (the LDS merge serves top-k <= 8)
    ("granular", 600, 20, "single"),     # ... with five tools per sample
    ("cloth", 4900, 5, "batch"),         # N = 4 901 > 4 352: bin_kernel and finalize_connect_kernel re-read the sample instead of caching it in registers
    ("cloth", 4900, 5, "single"),
])
def test_edges_connect_tools_all_beyond_the_fast_paths_vs_oracle(material, n_obj, topk, variant):
    """connect_tools_all = True on shapes that leave the register / LDS fast paths of the r05 edge builder (kept-sender lists longer than 8, samples
    larger than 17 x 256 slots): the edge lists still equal the oracle's bit for bit."""
    g = synth.make_graph_inputs(material, n_obj, 2, seed=33, **(dict(spacing=0.1) if material == "rope" else {}))
    radius = synth.MATERIALS[material]["radius"]
    pos = g["state"][:, -1].copy()
    n_rel, recv, send = ago.build_edges(pos, radius, g["mask"], g["tool_mask"], topk, True, variant)
    csr = aggraph.build_edges(t(pos), radius, t(g["mask"]), t(g["tool_mask"]), topk, True, variant, max_tools=g["n_tools"])
    assert csr.n_rel().cpu().tolist() == n_rel.tolist()
    for b, (r, s_) in enumerate(csr.to_lists()):
        assert np.array_equal(r, recv[b, :n_rel[b]]) and np.array_equal(s_, send[b, :n_rel[b]]), f"sample {b}"


@pytest.mark.parametrize("n_obj,seed", [(300, 1), (2000, 2), (5000, 3)])
def test_edges_topk20_packed_key_boundary_vs_oracle(n_obj, seed):
    """Top-k 20 (granular) takes the packed-key selection kernel (csrc/ag_edges.hip select_lanes_packed_kernel): 32-bit keys that
    keep only the leading bits of d.  Clouds built to make the K-th / (K+1)-th candidates (nearly) indistinguishable in those
    bits -- rings of ~45 senders at the same distance up to a few ulps around every centre, exact duplicates (equal d: the tie
    goes to the lower sender index), and a plain dense blob -- must still give the oracle's edge lists bit for bit."""
    rng = np.random.default_rng(seed)
    n_c = max(n_obj // 50, 1)
    centres = rng.uniform(0, 1.0 * np.sqrt(n_c), (n_c, 2))
    pts = []
    for c in centres:
        m = 45
        ang = rng.uniform(0, 2 * np.pi, m)
        rad = 0.3 * (1 + rng.integers(-3, 4, m) * 6e-8)              # a handful of ulps around r = 0.3
        pts.append(np.stack([c[0] + rad * np.cos(ang), np.zeros(m), c[1] + rad * np.sin(ang)], 1))
        pts.append(np.array([[c[0], 0.0, c[1]]]))
    pts = np.concatenate(pts)[:n_obj]
    if len(pts) < n_obj:
        pts = np.concatenate([pts, rng.uniform(0, 0.5, (n_obj - len(pts), 3)) * [1, 0.05, 1]])
    pts[n_obj // 3: n_obj // 3 + 20] = pts[0]                         # exact duplicates
    g = synth.make_graph_inputs("granular", n_obj, 2, seed=seed)
    pos = g["state"][:, -1].copy()
    pos[0, :n_obj] = pts.astype(np.float32)
    n_rel, recv, send = ago.build_edges(pos, 0.4, g["mask"], g["tool_mask"], 20, False, "batch")
    csr = aggraph.build_edges(t(pos), 0.4, t(g["mask"]), t(g["tool_mask"]), 20, False, "batch", max_tools=g["n_tools"])
    assert csr.n_rel().cpu().tolist() == n_rel.tolist()
    for b, (r, s_) in enumerate(csr.to_lists()):
        assert np.array_equal(r, recv[b, :n_rel[b]]) and np.array_equal(s_, send[b, :n_rel[b]]), f"sample {b}"


def test_edges_dropin_dense_signature():
    g = load_golden("edges_rope64_batch")
    Rr, Rs = aggraph.construct_edges_from_states_batch(t(g["pos"]), 0.5, t(g["mask"]), t(g["tool_mask"]), topk=10,
                                                       connect_tools_all=False)
    assert Rr.shape == (3, int(g["n_rel"].max()), 64) and Rr.dtype == torch.float32
    for b in range(3):
        n = g["n_rel"][b]
        assert torch.equal(Rr[b, :n].argmax(-1).cpu(), torch.from_numpy(g["recv"][b, :n]).long())
        assert torch.equal(Rs[b, :n].argmax(-1).cpu(), torch.from_numpy(g["send"][b, :n]).long())
        assert Rr[b, n:].abs().sum() == 0
    g = load_golden("edges_cloth256_single")
    Rr, Rs = aggraph.construct_edges_from_states(t(g["pos"][0]), 0.75, t(g["mask"][0]), t(g["tool_mask"][0]), topk=5,
                                                 connect_tools_all=True)
    assert Rr.shape == (int(g["n_rel"][0]), 257)
    assert torch.equal(Rs.argmax(-1).cpu(), torch.from_numpy(g["send"][0, :g["n_rel"][0]]).long())


# ------------------------------------------------------------------------------------------ forward
@pytest.mark.parametrize("name", golden_files("fwd_"))
def test_forward_golden(name, weights, prec):
    g = load_golden(name)
    material = str(g["material"])
    m = make_model(weights_for(g, weights), material, float(g["decoder_scale"]), prec)
    N = g["attrs"].shape[1]
    csr = csr_from_lists(g["n_rel"], g["recv"], g["send"], N)
    kw = {material + "_physics_param": t(g["phys"])}
    pos, mot = m(t(g["state"]), t(g["attrs"]), csr, None, t(g["p_instance"]), action=t(g["action"]), **kw)
    # the scaled-decoder clamp golden predicts motions of several hundred: pred_motion is held in relative terms there ...
    scale = max(1.0, float(np.abs(g["pred_motion"]).max()))
    assert np.abs(mot.cpu().numpy() - g["pred_motion"]).max() <= TOL_BY_PREC[prec] * scale
    # ... and pred_pos (model.py:309: state + clamp(motion, +-100)) absolutely wherever the reference's motion is clamped: the clamp removes
    # the deviation, what is left is the fp32 addition both sides perform; unclamped components carry the motion's deviation
    clamped = np.abs(g["pred_motion"]) >= 100.0 + 1.0
    dpos = np.abs(pos.cpu().numpy() - g["pred_pos"])
    if clamped.any():
        assert dpos[clamped].max() <= 1e-4
    assert dpos.max() <= TOL_BY_PREC[prec] * scale
    assert m.take_status() == 0            # no fp16 overflow / non-finite value on any golden, in any mode


def test_forward_dense_onehot_inputs_dropin(weights, model, prec):
    """model(**graph) with the reference's dict, dense Rr/Rs padded with all-zero rows (pad_torch, utils.py:37-46)."""
    g = load_golden("fwd_rope64")
    B, N = g["attrs"].shape[:2]
    E = int(g["n_rel"].max()) + 37
    Rr = torch.zeros(B, E, N, device=DEV)
    Rs = torch.zeros(B, E, N, device=DEV)
    for b in range(B):
        n = g["n_rel"][b]
        Rr[b, torch.arange(n), t(g["recv"][b, :n]).long()] = 1
        Rs[b, torch.arange(n), t(g["send"][b, :n]).long()] = 1
    graph = dict(state=t(g["state"]), attrs=t(g["attrs"]), Rr=Rr, Rs=Rs, p_instance=t(g["p_instance"]),
                 action=t(g["action"]), rope_physics_param=t(g["phys"]), obj_mask=None, p_rigid=torch.zeros(B, 1))
    pos, mot = model(**graph)
    assert np.abs(mot.cpu().numpy() - g["pred_motion"]).max() <= TOL_BY_PREC[prec]
    assert np.abs(pos.cpu().numpy() - g["pred_pos"]).max() <= TOL_BY_PREC[prec]


@pytest.mark.parametrize("material,n_obj,batch,kw", [
    ("rope", 1000, 2, dict(spacing=0.1)),
    ("granular", 2000, 1, {}),
    ("cloth", 4096, 1, {}),
    ("rope", 93, 5, dict(spacing=0.1, n_pad=7)),             # ragged: rows not a multiple of 32/128, padded slots
])
def test_forward_vs_oracle(material, n_obj, batch, kw, weights, prec):
    g = synth.make_graph_inputs(material, n_obj, batch, seed=4, **kw)
    mm = synth.MATERIALS[material]
    m = make_model(weights, material, prec=prec)
    pos_now = g["state"][:, -1]
    csr = aggraph.build_edges(t(pos_now), mm["radius"], t(g["mask"]), t(g["tool_mask"]), mm["topk"],
                              mm["connect_tools_all"], "batch", max_tools=g["n_tools"])
    n_rel, recv, send = ago.build_edges(pos_now, mm["radius"], g["mask"], g["tool_mask"], mm["topk"],
                                        mm["connect_tools_all"], "batch")
    ref_pos, ref_mot = ago.forward(weights, g["state"], g["attrs"], g["action"], g["p_instance"], g["phys"], n_rel, recv, send)
    kwp = {material + "_physics_param": t(g["phys"])}
    pos, mot = m(t(g["state"]), t(g["attrs"]), csr, None, t(g["p_instance"]), action=t(g["action"]), **kwp)
    assert np.abs(mot.cpu().numpy() - ref_mot).max() <= TOL_BY_PREC[prec]
    assert np.abs(pos.cpu().numpy() - ref_pos).max() <= TOL_FWD


def test_forward_no_edges_and_single_node(weights, model, prec):
    """Empty adjacency (all particles isolated and masked out of the graph) and a 1-particle cloud."""
    g = synth.make_graph_inputs("rope", 5, 2, seed=1, spacing=10.0)
    N = g["attrs"].shape[1]
    empty = aggraph.CSREdges(torch.zeros(2 * N + 1, dtype=torch.int32, device=DEV), torch.zeros(1, dtype=torch.int32, device=DEV),
                             torch.zeros(1, dtype=torch.int32, device=DEV), 2, N, 0)
    pos, mot = model(t(g["state"]), t(g["attrs"]), empty, None, t(g["p_instance"]), action=t(g["action"]),
                     rope_physics_param=t(g["phys"]))
    z = np.zeros((2, 1), np.int32)
    ref_pos, ref_mot = ago.forward(weights, g["state"], g["attrs"], g["action"], g["p_instance"], g["phys"],
                                   np.zeros(2, np.int32), z, z)
    assert np.abs(mot.cpu().numpy() - ref_mot).max() <= TOL_BY_PREC[prec]


def test_forward_is_batch_composition_independent(weights, model):
    """Sample b's output does not depend on what else is in the batch or where it sits (rows are independent;
    the MFMA chain per row is identical) -> bitwise equality between B=1 and a slot inside B=7."""
    g = synth.make_graph_inputs("rope", 300, 7, seed=8, spacing=0.1)
    kw = dict(rope_physics_param=t(g["phys"]))
    csr = aggraph.build_edges(t(g["state"][:, -1]), 0.5, t(g["mask"]), t(g["tool_mask"]), 10, False, "batch", max_tools=1)
    pos7, mot7 = model(t(g["state"]), t(g["attrs"]), csr, None, t(g["p_instance"]), action=t(g["action"]), **kw)
    b = 4
    one = {k: g[k][b:b + 1] for k in ("state", "attrs", "action", "p_instance", "phys", "mask", "tool_mask")}
    csr1 = aggraph.build_edges(t(one["state"][:, -1]), 0.5, t(one["mask"]), t(one["tool_mask"]), 10, False, "batch", max_tools=1)
    pos1, mot1 = model(t(one["state"]), t(one["attrs"]), csr1, None, t(one["p_instance"]), action=t(one["action"]),
                       rope_physics_param=t(one["phys"]))
    assert torch.equal(mot7[b], mot1[0]) and torch.equal(pos7[b], pos1[0])


def test_forward_repeatable_with_partial_last_tile(model):
    """Regression: a 128-row tile whose trailing waves hold no valid row must still drain its share of the weight
    DMA before the tile barrier (the counted s_waitcnt assumes every wave issued its epilogue stores).  E % 128 and
    B*N % 128 are small here, so waves 1-3 of the last tiles are empty; 40 repeats must be bit-identical."""
    g = synth.make_graph_inputs("rope", 300, 1, seed=9, spacing=0.1)
    csr = aggraph.build_edges(t(g["state"][:, -1]), 0.5, t(g["mask"]), t(g["tool_mask"]), 10, False, "batch", max_tools=1)
    n_edges = int(csr.row_ptr[-1].item())
    assert 0 < n_edges % 128 <= 64 and 0 < 301 % 128 <= 64
    args = (t(g["state"]), t(g["attrs"]), csr, None, t(g["p_instance"]))
    kw = dict(action=t(g["action"]), rope_physics_param=t(g["phys"]))
    _, first = model(*args, **kw)
    for _ in range(40):
        _, again = model(*args, **kw)
        assert torch.equal(first, again)


@pytest.mark.parametrize("material,n_obj,batch,kw", [
    ("rope", 700, 5, dict(spacing=0.1)),          # partial last 32-edge block, more blocks than workgroups
    ("rope", 40, 1, dict(spacing=0.1)),           # fewer blocks than the pipeline is deep
    ("granular", 2000, 2, {}),                    # saturated top-20 rows
    ("cloth", 1024, 3, {}),                       # connect_tools_all rows
])
def test_weight_stationary_edge_encoder_is_bitwise_the_streaming_kernel(weights, material, n_obj, batch, kw):
    """Precision mode 2 runs its edge stack on the weight-stationary kernel by default (weights in registers, 32-edge blocks
    pipelined through the eight waves of one workgroup per CU, inline-asm MFMAs); ag_set_option("edge_stationary", 0) selects
    the streaming kernel with the same two-product fp16 arithmetic.  Every row keeps its accumulation order, so the outputs
    must be bit-identical and repeatable — which also pins the hand-managed MFMA hazards and LDS hand-over of the new kernel."""
    m = make_model(weights, material, prec="fast")
    g = synth.make_graph_inputs(material, n_obj, batch, seed=17, **kw)
    mm = synth.MATERIALS[material]
    csr = aggraph.build_edges(t(g["state"][:, -1]), mm["radius"], t(g["mask"]), t(g["tool_mask"]), mm["topk"], mm["connect_tools_all"],
                              "batch", max_tools=g["n_tools"])
    args = (t(g["state"]), t(g["attrs"]), csr, None, t(g["p_instance"]))
    kw2 = {"action": t(g["action"]), material + "_physics_param": t(g["phys"])}
    m.set_option("edge_stationary", 0)
    _, ref = m(*args, **kw2)
    m.set_option("edge_stationary", 1)
    _, out = m(*args, **kw2)
    assert torch.isfinite(out).all() and torch.equal(ref, out)
    for _ in range(10):
        _, again = m(*args, **kw2)
        assert torch.equal(out, again)
    assert m.take_status() & 1 == 0



@pytest.mark.parametrize("mode", ["fast", "bf16x3"])
@pytest.mark.parametrize("material,n_obj,batch,dedup", [("rope", 700, 5, 2), ("rope", 40, 1, 2), ("granular", 2000, 2, 0), ("cloth", 1024, 3, 2), ("rope", 1000, 40, 1)])
def test_weight_stationary_node_update_is_bitwise_the_streaming_kernel(weights, mode, material, n_obj, batch, dedup):
    """The rounds before the last run their node update on the weight-stationary kernel by default (r05: one 256-thread workgroup per CU keeps the
    fifteen (layer, out-tile) units' split-bf16 weights in registers, 32-row blocks flow through LDS as a two-stage pipeline, inline-asm MFMAs);
    ag_set_option("node_stationary", 0) selects the streaming kernel.  Every accumulator keeps its order: bit-identical, repeatable — with a partial
    last 32-row block, fewer blocks than workgroups, compact (de-duplicated) and packed residual tables, and through a two-stream rollout."""
    m = make_model(weights, material, prec=mode)
    m.set_option("node_dedup", dedup)
    g = synth.make_graph_inputs(material, n_obj, batch, seed=19, **(dict(spacing=0.1) if material == "rope" else {}))
    mm = synth.MATERIALS[material]
    csr = aggraph.build_edges(t(g["state"][:, -1]), mm["radius"], t(g["mask"]), t(g["tool_mask"]), mm["topk"], mm["connect_tools_all"],
                              "batch", max_tools=g["n_tools"])
    args = (t(g["state"]), t(g["attrs"]), csr, None, t(g["p_instance"]))
    kw2 = {"action": t(g["action"]), material + "_physics_param": t(g["phys"])}
    m.set_option("node_stationary", 0)
    pos0, ref = m(*args, **kw2)
    m.set_option("node_stationary", 1)
    for _ in range(5):
        pos1, out = m(*args, **kw2)
        assert torch.isfinite(out).all() and torch.equal(ref, out) and torch.equal(pos0, pos1)
    if material == "rope" and n_obj == 1000:
        state, act = synth.make_mpc_inputs("rope", 300, 24, seed=6, len_lo=3, len_hi=5.9, spacing=0.1)
        on = dynamics(t(state), t(act), m, DEV, _ppm("rope"))["state_seqs"]
        m.set_option("node_stationary", 0)
        assert torch.equal(on, dynamics(t(state), t(act), m, DEV, _ppm("rope"))["state_seqs"])
    assert m.take_status() == 0


def test_fused_segment_reduce_equals_the_separate_kernel(weights):
    """ag_set_option("fuse_aggregate", 2): the round's segment reduce inside node_update (no `agg` table, no aggregate launch) adds
    every node's messages in the same order as aggregate_half_kernel, so outputs are bit-identical to the default path — on a
    batch with partial row tiles, isolated nodes and a node-count that is not a multiple of 32; other precision modes ignore it."""
    m = make_model(weights, prec="fast")
    g = synth.make_graph_inputs("rope", 333, 3, seed=5, spacing=0.1)
    g["mask"][1, 40:60] = False                                   # a gap: receivers without edges inside a row tile
    csr = aggraph.build_edges(t(g["state"][:, -1]), 0.5, t(g["mask"]), t(g["tool_mask"]), 10, False, "batch", max_tools=1)
    args = (t(g["state"]), t(g["attrs"]), csr, None, t(g["p_instance"]))
    kw = dict(action=t(g["action"]), rope_physics_param=t(g["phys"]))
    _, ref = m(*args, **kw)
    m.set_option("fuse_aggregate", 2)
    _, got = m(*args, **kw)
    assert torch.isfinite(got).all() and torch.equal(ref, got)
    for _ in range(10):
        assert torch.equal(m(*args, **kw)[1], got)
    state, act = synth.make_mpc_inputs("rope", 300, 20, seed=4, len_lo=3, len_hi=4.9, spacing=0.1)     # and through the 2-stream rollout
    fused = dynamics(t(state), t(act), m, DEV, _ppm("rope"))["state_seqs"]
    m.set_option("fuse_aggregate", 0)
    assert torch.equal(fused, dynamics(t(state), t(act), m, DEV, _ppm("rope"))["state_seqs"])
    m.set_option("fuse_aggregate", 2)
    m.set_option("precision", 1)                                  # fp32 per-edge table: the option falls back to the separate launch
    _, p1 = m(*args, **kw)
    m.set_option("fuse_aggregate", 0)
    assert torch.equal(m(*args, **kw)[1], p1)


@pytest.mark.parametrize("case", ["rope_gap", "distinct_actions", "granular_tools", "cloth_padded", "ten_classes"])
def test_node_encoder_deduplication_is_bitwise_the_per_node_encoder(weights, prec, case):
    """ag_set_option("node_dedup", 1) (default): particle_encode / Pn / the first round's Hr, Hs are computed once per distinct
    [attrs | phys | action] row of a sample and read through an index (csrc/ag_mlp.hip node_classify_kernel).  Outputs must equal the
    per-node encoder bit for bit — with the usual two classes (objects, tool), with a private row for EVERY node (distinct actions),
    with several tools, padded slots, and more distinct rows than the shared slots per sample."""
    mat = {"rope_gap": "rope", "distinct_actions": "rope", "granular_tools": "granular", "cloth_padded": "cloth", "ten_classes": "rope"}[case]
    n_obj = {"rope_gap": 333, "distinct_actions": 150, "granular_tools": 500, "cloth_padded": 256, "ten_classes": 200}[case]
    kw = dict(spacing=0.1) if mat == "rope" else {}
    g = synth.make_graph_inputs(mat, n_obj, 3, seed=8, n_pad=9 if case == "cloth_padded" else 0, **kw)
    rng = np.random.default_rng(4)
    if case == "rope_gap":
        g["mask"][1, 40:60] = False
    if case == "distinct_actions":
        g["action"] = rng.normal(0, 0.1, g["action"].shape).astype(np.float32)
    if case == "ten_classes":
        g["action"][:, :n_obj] = (rng.integers(0, 10, (3, n_obj, 1)) * 0.01).astype(np.float32)
    if case == "granular_tools":
        g["action"][:, -3:] = rng.normal(0, 0.1, (3, 3, 3)).astype(np.float32)         # tools that do not share a row
    mm = synth.MATERIALS[mat]
    m = make_model(weights, mat, prec=prec)
    csr = aggraph.build_edges(t(g["state"][:, -1]), mm["radius"], t(g["mask"]), t(g["tool_mask"]), mm["topk"], mm["connect_tools_all"], "batch",
                              max_tools=g["n_tools"])
    args = (t(g["state"]), t(g["attrs"]), csr, None, t(g["p_instance"]))
    kwp = {"action": t(g["action"]), mat + "_physics_param": t(g["phys"] * np.array([[1.0], [0.5], [0.25]], np.float32))}
    m.set_option("node_dedup", 0)
    pos0, mot0 = m(*args, **kwp)
    m.set_option("node_dedup", 2)                                 # 2 = always (1, the default, skips it for launches this small)
    for _ in range(3):
        pos1, mot1 = m(*args, **kwp)
        assert torch.isfinite(mot1).all() and torch.equal(mot0, mot1) and torch.equal(pos0, pos1)
    if prec == "fast":                                            # and with the reduce fused into node_update, and through a rollout
        m.set_option("fuse_aggregate", 2)
        assert torch.equal(m(*args, **kwp)[1], mot0)
        m.set_option("fuse_aggregate", 0)
    if case == "rope_gap":
        state, act = synth.make_mpc_inputs("rope", 300, 20, seed=4, len_lo=3, len_hi=4.9, spacing=0.1)
        on = dynamics(t(state), t(act), m, DEV, _ppm("rope"))["state_seqs"]
        m.set_option("node_dedup", 0)
        assert torch.equal(on, dynamics(t(state), t(act), m, DEV, _ppm("rope"))["state_seqs"])


@pytest.mark.parametrize("mode,fuse", [("fast", 0), ("fast", 2), ("bf16x3", 0), ("f32", 0)])      # (the fused reduce exists in the default mode only)
def test_dedup_overflow_falls_back_to_the_per_node_encoder(weights, mode, fuse):
    """The compact tables of the de-duplicated node encoder are bounded (8 shared rows per sample + ~B N / 16 private rows, at least 1 024).
    Inputs with more distinct rows — here a private row for every one of 3 x 6 001 nodes — raise a device flag and the call runs through the
    per-node encoder (node_classify_kernel -> ovf; every consumer tests it): same bits as node_dedup 0, in one forward and through a rollout
    whose later steps re-encode from the flag alone."""
    n_obj, B = 6000, 3
    g = synth.make_graph_inputs("rope", n_obj, B, seed=8, spacing=0.1)
    rng = np.random.default_rng(5)
    g["action"] = rng.normal(0, 0.05, g["action"].shape).astype(np.float32)
    assert B * (n_obj + 1) > max(B * (n_obj + 1) // 16, 1024) + 8 * B
    m = make_model(weights, "rope", prec=mode)
    m.set_option("fuse_aggregate", fuse)
    csr = aggraph.build_edges(t(g["state"][:, -1]), 0.5, t(g["mask"]), t(g["tool_mask"]), 10, False, "batch", max_tools=1)
    args = (t(g["state"]), t(g["attrs"]), csr, None, t(g["p_instance"]))
    kwp = {"action": t(g["action"]), "rope_physics_param": t(g["phys"])}
    thr = aggraph.threshold_sq(0.5, B, torch.device(DEV), _lib.AG_VARIANT_BATCH)
    rep = torch.full((B,), 3, dtype=torch.int32, device=DEV)

    def both():
        pos, mot = m(*args, **kwp)
        seq = rollout(m, t(g["state"]), t(g["action"]), t(g["attrs"]), t(g["p_instance"]), t(g["phys"]), t(g["mask"]), t(g["tool_mask"]), thr, rep, 3,
                      10, False, 1)
        return pos, mot, seq

    m.set_option("node_dedup", 0)
    ref = both()
    m.set_option("node_dedup", 2)
    for _ in range(2):
        got = both()
        assert all(torch.isfinite(a).all() and torch.equal(a, b) for a, b in zip(got, ref))
    assert m.take_status() == 0


@pytest.mark.parametrize("mode", ["fast", "bf16x3", "f32"])
@pytest.mark.parametrize("material,n_obj,batch,steps", [("rope", 1000, 17, 3), ("granular", 2000, 3, 2), ("cloth", 1024, 4, 3), ("rope", 100, 5, 3)])
def test_self_edge_elision_is_bitwise_the_full_pipeline(weights, mode, material, n_obj, batch, steps):
    """ag_set_option("self_edges", 1) (default, r06): ag_rollout's edge builder leaves the self-loops of attribute classes (1, 0) / (0, 1) out of the
    COO list and the per-edge table, the edge encoder computes one row per class (synthetic edges behind the list) and the segment reduce adds it at
    the self-loop's position in the receiver's order — the same bits as every edge through the pipeline (option 0), in every arithmetic mode, with the
    weight-stationary and the streaming edge encoder, the fused reduce, without node de-duplication, on two streams, with connect_tools_all (cloth:
    the tool's own self-loop), with 5 tools (granular), on the brute-force edge path (N < 256) and through dynamics_masked's invalid slots."""
    m = make_model(weights, material, prec=mode)
    kw = dict(spacing=0.1) if material == "rope" else {}
    state, act = synth.make_mpc_inputs(material, n_obj, batch, seed=11, len_lo=steps, len_hi=steps + 0.9, **kw)

    def run():
        return dynamics(t(state), t(act), m, DEV, _ppm(material))["state_seqs"].clone()

    m.set_option("self_edges", 0)
    ref = run()
    m.set_option("self_edges", 1)
    assert torch.isfinite(ref).all() and torch.equal(ref, run())
    variants = [("node_dedup", 0, 1), ("rollout_streams", 2, 0)]
    if mode == "fast":
        variants += [("edge_stationary", 0, 1), ("fuse_aggregate", 2, 0)]
    for name, val, back in variants:
        m.set_option(name, val)
        assert torch.equal(ref, run()), name
        m.set_option(name, back)
    if material == "rope" and n_obj == 1000:      # invalid object slots (no self-loop there), per-sample masks
        rng = np.random.default_rng(2)
        st = np.repeat(state[None], 4, 0) + rng.normal(0, 0.002, (4,) + state.shape).astype(np.float32)
        sm = rng.uniform(size=(4, n_obj)) < 0.8
        a4 = act[:4, 0]
        got = dynamics_masked(t(st), t(sm), t(a4), m, DEV, _ppm(material))["state_seqs"].clone()
        m.set_option("self_edges", 0)
        assert torch.equal(got, dynamics_masked(t(st), t(sm), t(a4), m, DEV, _ppm(material))["state_seqs"])
    assert m.take_status() == 0


@pytest.mark.parametrize("mode", ["fast", "f32"])
def test_self_edge_elision_keeps_other_attribute_pairs_and_duplicate_particles(weights, mode):
    """Only the attribute pairs (1, 0) and (0, 1) have a class row: a node with any other pair keeps its self-loop as a real edge.  And a self-loop
    that the top-k rule drops (more than top-k exact duplicates of a particle with lower indices) is not there to elide.  Same bits as option 0."""
    m = make_model(weights, "rope", prec=mode)
    g = synth.make_graph_inputs("rope", 600, 3, seed=31, spacing=0.1)
    rng = np.random.default_rng(4)
    odd = rng.uniform(size=g["attrs"].shape[:2]) < 0.3
    g["attrs"][odd] = np.array([0.5, 0.25], np.float32)
    g["attrs"][0, 5] = np.array([1.0, 1.0], np.float32)
    g["state"][1, :, 100:130] = g["state"][1, :, 100:101]          # 30 exact duplicates: top-k (10) keeps the ten lowest indices for every one of them
    thr = aggraph.threshold_sq(0.5, 3, torch.device(DEV), _lib.AG_VARIANT_BATCH)
    rep = torch.full((3,), 3, dtype=torch.int32, device=DEV)

    def run():
        return rollout(m, t(g["state"]), t(g["action"]), t(g["attrs"]), t(g["p_instance"]), t(g["phys"]), t(g["mask"]), t(g["tool_mask"]), thr, rep, 3,
                       10, False, 1).clone()

    m.set_option("node_dedup", 2)
    m.set_option("self_edges", 0)
    ref = run()
    m.set_option("self_edges", 1)
    assert torch.isfinite(ref).all() and torch.equal(ref, run())
    assert m.take_status() == 0


@pytest.mark.parametrize("name", golden_files("fwd_"))
def test_agg_q16_option_holds_the_goldens(name, weights):
    """ag_set_option("agg_q16", 1) (r06; the default of precision mode 2): the segment reduce stores `agg` as q16 rows (the per-edge table's row layout: 320 B,
    one power-of-two scale per 32 features; unsigned values — the sums are >= 0: ag_q16_encode_segment) and node_update decodes them — one more 16-bit
    rounding per node and round, half the bytes of that table.  Every forward golden stays inside the mode's tolerance with status 0 with the option on
    and off; the streaming node_update and the reduce fused into node_update apply the same rounding (same bits); the other modes ignore the option."""
    g = load_golden(name)
    material = str(g["material"])
    N = g["attrs"].shape[1]
    csr = csr_from_lists(g["n_rel"], g["recv"], g["send"], N)
    kw = {material + "_physics_param": t(g["phys"])}
    scale = max(1.0, float(np.abs(g["pred_motion"]).max()))
    m = make_model(weights_for(g, weights), material, float(g["decoder_scale"]), "fast")
    args = (t(g["state"]), t(g["attrs"]), csr, None, t(g["p_instance"]))
    assert m.get_option("agg_q16") == 1
    _, mot = m(*args, action=t(g["action"]), **kw)
    mot = mot.clone()
    assert np.abs(mot.cpu().numpy() - g["pred_motion"]).max() <= TOL_BY_PREC["fast"] * scale
    for name_, val in (("node_stationary", 0), ("fuse_aggregate", 2)):
        m.set_option(name_, val)
        _, mot1 = m(*args, action=t(g["action"]), **kw)
        assert torch.equal(mot1, mot), name_
    m.set_option("node_stationary", 1).set_option("fuse_aggregate", 0)
    m.set_option("agg_q16", 0)
    _, mot0 = m(*args, action=t(g["action"]), **kw)
    assert not torch.equal(mot, mot0)                  # the option is live
    assert np.abs(mot0.cpu().numpy() - g["pred_motion"]).max() <= TOL_BY_PREC["fast"] * scale
    assert m.take_status() == 0
    m2 = make_model(weights_for(g, weights), material, float(g["decoder_scale"]), "bf16x3")
    _, ref = m2(*args, action=t(g["action"]), **kw)
    ref = ref.clone()
    m2.set_option("agg_q16", 0)
    _, got = m2(*args, action=t(g["action"]), **kw)
    assert torch.equal(ref, got)


@pytest.mark.parametrize("material,n_obj,batch,steps", [("rope", 1000, 48, 5), ("rope", 333, 7, 4), ("cloth", 1024, 9, 3), ("granular", 2000, 6, 3)])
def test_agg_q16_rollouts_keep_their_bitwise_equalities(weights, material, n_obj, batch, steps):
    """With `agg` as q16 rows the sums the reduce forms are unchanged and their encoding is a function of the sum alone: self-edge elision, node
    de-duplication, the shared-state rollout and a rollout alone vs inside a batch still agree bit for bit (row counts that are not a multiple of 32, an
    scratch workspace full of 0xFF bytes), and the rollout stays within 10x the one-step tolerance of the default arithmetic after `steps` steps."""
    m = make_model(weights, material, prec="fast")
    kw = dict(spacing=0.1) if material == "rope" else {}
    state, act = synth.make_mpc_inputs(material, n_obj, batch, n_look=1, seed=29, len_lo=steps, len_hi=steps + 0.9, **kw)

    def run(sl=slice(None)):
        return dynamics(t(state), t(act[sl]), m, DEV, _ppm(material))["state_seqs"].clone()

    m.set_option("agg_q16", 0)
    base = run()
    m.set_option("agg_q16", 1)
    ref = run()
    torch.cuda.synchronize()
    for buf in aggraph._WS.values():
        buf.fill_(0xFF)
    assert torch.isfinite(ref).all() and torch.equal(ref, run())
    # (a rollout re-builds its edges from its own predictions: a top-k near-tie resolved the other way moves a few particles by more — the drift every
    # arithmetic mode shows against the reference, profiles/r06_rollout_drift.txt — so the bound is on all but 1 % of the coordinates)
    assert ((ref - base).abs() > 10 * TOL_BY_PREC["fast"]).float().mean().item() < 0.01
    assert torch.equal(ref[:3], run(slice(0, 3)))
    for name, val, back in [("self_edges", 0, 1), ("node_dedup", 0, 1), ("shared_state", 1, 0), ("rollout_streams", 2, 0), ("node_stationary", 0, 1),
                            ("fuse_aggregate", 2, 0)]:
        m.set_option(name, val)
        assert torch.equal(ref, run()), name
        m.set_option(name, back)
    assert m.take_status() == 0


@pytest.mark.parametrize("mode,material,n_obj,batch,steps", [
    ("fast", "rope", 1000, 96, 6), ("bf16x3", "rope", 1000, 40, 4), ("f32", "rope", 300, 33, 5), ("fast", "granular", 2000, 12, 3),
    ("fast", "cloth", 1024, 9, 4), ("fast", "rope", 100, 7, 4)])
def test_shared_state_rollout_is_bitwise_the_plain_rollout(weights, mode, material, n_obj, batch, steps):
    """ag_set_option("shared_state", 1) (r06): dynamics() rolls ONE cloud out under `batch` sampled pushes, so the engine rolls the tool-less base
    trajectory out once and computes per sample only the rows that can differ from it (dirty nodes, rows with another edge list or a dirty sender,
    3-hop closure per model step); every other particle takes the base's prediction.  Same bits as the plain rollout — state_seqs of dynamics() in
    every arithmetic mode, with and without self-edge elision, the streaming kernels, a multi-step look-ahead (per-sample states from the second
    look-ahead step on: everything private), connect_tools_all (cloth: a touching tool makes the whole sample private)."""
    m = make_model(weights, material, prec=mode)
    kw = dict(spacing=0.1) if material == "rope" else {}
    state, act = synth.make_mpc_inputs(material, n_obj, batch, n_look=2 if n_obj <= 300 else 1, seed=13, len_lo=steps, len_hi=steps + 0.9, **kw)

    def run():
        return dynamics(t(state), t(act), m, DEV, _ppm(material))["state_seqs"].clone()

    ref = run()
    m.set_option("shared_state", 1)
    for fill in (0xFF, 0x00, None):      # (whatever the scratch memory held: skipped samples leave their part of the graph arrays unwritten)
        run()                            # sizes the workspace for the option
        torch.cuda.synchronize()
        for buf in aggraph._WS.values():
            if fill is None:
                buf.random_(0, 256)
            else:
                buf.fill_(fill)
        assert torch.isfinite(ref).all() and torch.equal(ref, run())
    for name, val, back in [("self_edges", 0, 1), ("node_stationary", 0, 1), ("node_dedup", 0, 1)] + ([("edge_stationary", 0, 1)] if mode == "fast" else []):
        m.set_option(name, val)
        assert torch.equal(ref, run()), name
        m.set_option(name, back)
    assert m.take_status() == 0


def test_shared_state_rollout_with_distinct_states_degrades_to_the_plain_rollout(weights):
    """Per-sample DISTINCT states (dynamics_masked: every sample its own cloud and mask; and ag_rollout with perturbed per-sample inputs): nothing can be
    shared — every node is dirty from the first step on and the option reduces to the plain rollout plus one base sample.  Same bits.  Partially
    shared inputs too: half of the samples share sample 0's cloud, a quarter have another physics parameter, some a displaced particle."""
    m = make_model(weights, "rope", prec="fast")
    rng = np.random.default_rng(6)
    state, act = synth.make_mpc_inputs("rope", 400, 12, seed=5, len_lo=4, len_hi=4.9, spacing=0.1)
    st = np.repeat(state[None], 12, 0) + rng.normal(0, 0.002, (12,) + state.shape).astype(np.float32)
    sm = rng.uniform(size=(12, 400)) < 0.85
    ref = dynamics_masked(t(st), t(sm), t(act[:, 0]), m, DEV, _ppm("rope"))["state_seqs"].clone()
    m.set_option("shared_state", 1)
    assert torch.equal(ref, dynamics_masked(t(st), t(sm), t(act[:, 0]), m, DEV, _ppm("rope"))["state_seqs"])
    m.set_option("shared_state", 0)
    B = 16
    g = synth.make_graph_inputs("rope", 600, B, seed=9, spacing=0.1)
    g["state"][:] = g["state"][:1]                                   # one cloud ...
    g["state"][B // 2:, :, :600] += rng.normal(0, 0.003, (B - B // 2, 4, 600, 3)).astype(np.float32)      # ... but half of the samples have their own
    g["state"][3, :, 77] += 0.05                                     # one displaced particle
    g["state"][:, :, 600:] += rng.normal(0, 0.3, (B, 1, 1, 3)).astype(np.float32)                         # every sample its own tool position
    g["phys"][B // 4:B // 2] = 0.8
    g["action"][:, 600:] = rng.normal(0, 0.1, (B, 1, 3)).astype(np.float32)
    thr = aggraph.threshold_sq(0.5, B, torch.device(DEV), _lib.AG_VARIANT_BATCH)
    rep = t(rng.integers(1, 6, B).astype(np.int32))

    def run():
        return tuple(x.clone() for x in rollout(m, t(g["state"]), t(g["action"]), t(g["attrs"]), t(g["p_instance"]), t(g["phys"]), t(g["mask"]),
                                                t(g["tool_mask"]), thr, rep, 5, 10, False, 1, return_state=True))

    ref = run()
    m.set_option("shared_state", 1)
    got = run()
    assert all(torch.isfinite(a).all() and torch.equal(a, b) for a, b in zip(got, ref))
    assert m.take_status() == 0


@pytest.mark.parametrize("seed", range(12))
def test_shared_state_rollout_random_cases(weights, seed):
    """Randomised plain-vs-shared comparison through the raw rollout: material, 1 ... 700 particles (both edge-builder paths), 2 ... 40 samples, 1 ... 4 model
    steps, per-sample step counts, tools near / far / absent-minded (some samples park theirs far away), invalid object slots — shared by all samples or per
    sample —, padded slots, a few samples with a perturbed cloud, another physics parameter or radius, the masked-mean tool height, every arithmetic mode.
    Final states and recorded predictions equal bit for bit, status 0."""
    rng = np.random.default_rng(1000 + seed)
    material = ("rope", "granular", "cloth")[seed % 3]
    mm = synth.MATERIALS[material]
    n_obj = int(rng.choice([1, 7, 40, 150, 300, 700])) if material != "cloth" else int(rng.choice([16, 144, 625]))
    B, steps = int(rng.integers(2, 41)), int(rng.integers(1, 5))
    mode = ("fast", "bf16x3", "f32")[(seed // 3) % 3]
    m = make_model(weights, material, prec=mode)
    g = synth.make_graph_inputs(material, n_obj, B, seed=seed, n_pad=int(rng.integers(0, 3)), **(dict(spacing=0.1) if material == "rope" else {}))
    n_p, N, n_t = g["n_p"], g["attrs"].shape[1], g["n_tools"]
    g["state"][:] = g["state"][:1]                                           # one cloud ...
    far = rng.uniform(size=B) < 0.4                                          # ... every sample its own tool pose, some far away
    g["state"][:, :, n_p:] += rng.normal(0, 0.4, (B, 1, 1, 3)).astype(np.float32) + far[:, None, None, None] * np.float32(30.0)
    g["action"][:, n_p:] = rng.normal(0, 0.1, (B, 1, 3)).astype(np.float32)
    if rng.uniform() < 0.5:
        g["mask"][:, :n_obj] &= rng.uniform(size=(1, n_obj)) < 0.9           # invalid slots shared by all samples
    if rng.uniform() < 0.4:
        g["mask"][:, :n_obj] &= rng.uniform(size=(B, n_obj)) < 0.97          # ... and per sample (those nodes differ from the base)
    for b in rng.choice(B, size=max(1, B // 6), replace=False):              # a few samples that share less
        kind = rng.integers(0, 3)
        if kind == 0:
            g["state"][b, :, :n_obj] += rng.normal(0, 0.004, (4, n_obj, 3)).astype(np.float32)
        elif kind == 1:
            g["phys"][b] = 0.9
        else:
            g["state"][b, -1, int(rng.integers(0, n_obj))] += 0.03
    radius = np.full(B, mm["radius"], np.float32)
    if rng.uniform() < 0.3:
        radius[int(rng.integers(0, B))] *= 1.1
    thr = aggraph.threshold_sq(t(radius), B, torch.device(DEV), _lib.AG_VARIANT_BATCH)
    rep = t(rng.integers(1, steps + 1, B).astype(np.int32))
    masked = bool(rng.uniform() < 0.3)
    obj_mask = t(g["mask"][:, :n_p]) if masked else None
    if masked and not g["mask"][:, :n_p].any(1).all():
        masked, obj_mask = False, None

    def run():
        return tuple(x.clone() for x in rollout(m, t(g["state"]), t(g["action"]), t(g["attrs"]), t(g["p_instance"]), t(g["phys"]), t(g["mask"]), t(g["tool_mask"]),
                                                thr, rep, steps, mm["topk"], mm["connect_tools_all"], n_t,
                                                _lib.AG_HEIGHT_MASKED_MEAN if masked else _lib.AG_HEIGHT_MIN, obj_mask, 0.0, return_state=True))

    ref = run()
    m.set_option("shared_state", 1)
    run()
    torch.cuda.synchronize()
    for buf in aggraph._WS.values():
        buf.random_(0, 256)
    got = run()
    assert all(torch.equal(a, b) for a, b in zip(got, ref)), (material, n_obj, B, steps, mode)
    m.take_status()


def test_shared_state_rollout_at_the_mpc_shape_computes_a_fraction_of_the_edges(weights):
    """BASELINE configs[4] on one GPU (rope-1000 + tool, 1 024 sampled pushes x 15 model steps): bit-equal to the plain rollout, and the edge encoder
    walks a small fraction of the plain rollout's edges (the library's per-launch edge counter): most samples' tools never touch the rope."""
    import ctypes
    m = make_model(weights, "rope", prec="fast")
    state, act = synth.make_mpc_inputs("rope", 1000, 1024, seed=0, len_lo=15, len_hi=15.4, spacing=0.1)
    L, h = _lib.lib(), m.handle(torch.device(DEV))

    def run():
        _lib.check(L.ag_profile_enable(h, 1), "ag_profile_enable")
        out = dynamics(t(state), t(act), m, DEV, _ppm("rope"))["state_seqs"].clone()
        ms, cnt, edges = (ctypes.c_double * 6)(), (ctypes.c_int64 * 6)(), ctypes.c_int64()
        _lib.check(L.ag_profile_read(h, ms, cnt, ctypes.byref(edges)), "ag_profile_read")
        _lib.check(L.ag_profile_enable(h, 0), "ag_profile_enable")
        return out, edges.value

    ref, e_full = run()
    m.set_option("shared_state", 1)
    got, e_shared = run()
    assert torch.equal(ref, got) and m.take_status() == 0
    assert 0 < e_shared < 0.25 * e_full, (e_shared, e_full)


@pytest.mark.parametrize("mode", ["fast", "bf16x3"])
@pytest.mark.parametrize("node_ws", [1, 0])
def test_stale_workspace_bytes_never_reach_the_results_or_the_status(weights, mode, node_ws):
    """the weight-stationary node_update read `agg` rows past B*N when B*N is not a multiple of 32 — rows the segment
    reduce never writes, i.e. whatever the grow-only workspace held (a NaN there raised a spurious AG_STATUS through the q16 store of the padded
    Hs rows).  Fill every scratch buffer with 0xFF bytes (NaN as fp32, -1 as int32) in front of a forward and a bsz = 1 rollout at N = 1001:
    same bits as before the fill, status 0."""
    m = make_model(weights, "rope", prec=mode)
    m.set_option("node_stationary", node_ws)
    m.set_option("node_dedup", 2)
    g = synth.make_graph_inputs("rope", 1000, 1, seed=23, spacing=0.1)
    assert g["attrs"].shape[1] % 32 != 0
    csr = aggraph.build_edges(t(g["state"][:, -1]), 0.5, t(g["mask"]), t(g["tool_mask"]), 10, False, "batch", max_tools=1)
    args = (t(g["state"]), t(g["attrs"]), csr, None, t(g["p_instance"]))
    kwp = {"action": t(g["action"]), "rope_physics_param": t(g["phys"])}
    state, act = synth.make_mpc_inputs("rope", 1000, 1, seed=3, len_lo=3, len_hi=3.9, spacing=0.1)

    def both():
        pos, mot = m(*args, **kwp)
        return pos.clone(), mot.clone(), dynamics(t(state), t(act), m, DEV, _ppm("rope"))["state_seqs"].clone()

    ref = both()
    assert m.take_status() == 0
    for _ in range(2):
        torch.cuda.synchronize()
        for buf in aggraph._WS.values():
            buf.fill_(0xFF)
        got = both()
        assert all(torch.isfinite(a).all() and torch.equal(a, b) for a, b in zip(got, ref))
        assert m.take_status() == 0
    # and the same model right afterwards on inputs that DO fit: the flag is per call
    g2 = synth.make_graph_inputs("rope", 300, 2, seed=1, spacing=0.1)
    csr2 = aggraph.build_edges(t(g2["state"][:, -1]), 0.5, t(g2["mask"]), t(g2["tool_mask"]), 10, False, "batch", max_tools=1)
    a2 = (t(g2["state"]), t(g2["attrs"]), csr2, None, t(g2["p_instance"]))
    k2 = {"action": t(g2["action"]), "rope_physics_param": t(g2["phys"])}
    on = m(*a2, **k2)[1]
    m.set_option("node_dedup", 0)
    assert torch.equal(on, m(*a2, **k2)[1])


def test_node_deduplication_through_the_masked_rollout(weights):
    """dynamics_masked (sys-id driver): per-sample initial clouds with padded object slots (attrs (0, 0): a second object class), the
    node encoder hoisted out of the step loop — de-duplicated, forced and off must agree bit for bit, and with the golden."""
    g = load_golden("dynmask_rope40")
    m = make_model(weights, "rope", prec="fast")
    outs = []
    for dd in (0, 2, 1):
        m.set_option("node_dedup", dd)
        outs.append(dynamics_masked(t(g["state_init"]), t(g["state_mask"]), t(g["action"]), m, DEV, _ppm("rope"))["state_seqs"])
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    err = (np.abs(outs[0].cpu().numpy() - g["state_seqs"]) * g["state_mask"][..., None]).reshape(g["state_seqs"].shape[0], -1).max(1)
    for b in np.nonzero(err > TOL_FWD)[0]:      # every sample inside the gate, or a PROVEN top-k near-tie (as in test_dynamics_masked_golden; r05 review weak 1c: was a flat 1e-3)
        step, dev, gap = explain_divergence(m, weights, "rope", g["state_init"], g["action"], int(b), state_mask=g["state_mask"])
        print(f"dynmask_rope40[{b}] (fast, de-duplicated): top-k near-tie at step {step}: candidates {gap:.2e} apart, forward deviation {dev:.2e}")


def test_forward_translation_invariance(model):
    """Positions enter only through differences (model.py:168-173 skipped, :250): shifting the cloud by a
    power-of-two offset (exact in fp32 at this magnitude) leaves pred_motion unchanged to rounding."""
    g = synth.make_graph_inputs("rope", 200, 2, seed=2, spacing=0.1)
    csr = aggraph.build_edges(t(g["state"][:, -1]), 0.5, t(g["mask"]), t(g["tool_mask"]), 10, False, "batch", max_tools=1)
    args = (t(g["attrs"]), csr, None, t(g["p_instance"]))
    kw = dict(action=t(g["action"]), rope_physics_param=t(g["phys"]))
    _, m0 = model(t(g["state"]), *args, **kw)
    shift = torch.tensor([4.0, -2.0, 8.0], device=DEV)
    p1, m1 = model(t(g["state"]) + shift, *args, **kw)
    assert (m0 - m1).abs().max().item() <= 4e-5


def test_forward_permutation_equivariance(model, prec):
    """Relabelling the object particles permutes the outputs and changes nothing else (permutation
    invariance).  The edge SET is the same up to relabelling; only the order in which a receiver's messages are summed
    changes, so equality is to rounding, and the rebuilt edge lists are compared exactly."""
    g = synth.make_graph_inputs("rope", 150, 2, seed=4, spacing=0.1)
    n_p = g["p_instance"].shape[1]
    rng = np.random.default_rng(0)
    perm = np.concatenate([rng.permutation(n_p), np.arange(n_p, g["attrs"].shape[1])])      # tools stay in the trailing slots
    inv = np.argsort(perm)
    kw = dict(rope_physics_param=t(g["phys"]))

    def run(state, attrs, action, p_instance, mask, tool):
        csr = aggraph.build_edges(t(state[:, -1]), 0.5, t(mask), t(tool), 10, False, "batch", max_tools=1)
        pos, mot = model(t(state), t(attrs), csr, None, t(p_instance), action=t(action), **kw)
        return csr, pos.cpu().numpy(), mot.cpu().numpy()

    csr0, pos0, mot0 = run(g["state"], g["attrs"], g["action"], g["p_instance"], g["mask"], g["tool_mask"])
    csr1, pos1, mot1 = run(g["state"][:, :, perm], g["attrs"][:, perm], g["action"][:, perm], g["p_instance"][:, perm[:n_p]],
                           g["mask"][:, perm], g["tool_mask"][:, perm])
    for (r0, s0), (r1, s1) in zip(csr0.to_lists(), csr1.to_lists()):
        assert sorted(zip(r0.tolist(), s0.tolist())) == sorted(zip(perm[r1].tolist(), perm[s1].tolist()))
    tol = 2 * TOL_BY_PREC[prec]
    assert np.abs(mot1 - mot0[:, perm[:n_p]]).max() <= tol and np.abs(pos1 - pos0[:, perm[:n_p]]).max() <= tol
    assert inv[perm[3]] == 3


# ------------------------------------------------------------------------------------------ rollout drivers
# Multi-step rollouts rebuild the radius/top-k graph from PREDICTED positions every step, so arithmetic that is not
# bit-identical to the reference's (any GPU, any summation order) can flip an edge whose two candidate distances
# differ by less than the forward deviation; from that step on the trajectories differ by O(1e-2) (SURVEY.md §7
# H3/H4: "gate parity per step on identical graphs; report rollout drift separately").  Exact-fp32 mode matches
# every golden rollout; the split-bf16 modes (forward deviation ~5e-6) flip one neighbour in one granular sample whose
# top-20 lists are saturated.  There is no waiver list: a sample outside the gate must be PROVEN to be such a near-tie
# (explain_divergence), otherwise the test fails.


def _ppm(material):
    ppm = configs.ppm_optimizer_stub(material)
    ppm.physics_param = {material: torch.tensor([0.5], device=DEV)}
    return ppm


def explain_divergence(model, weights, material, state, action, b, state_mask=None, allow_drift=False):
    """Sample b of dynamics(state, action) — or, with `state_mask`, of dynamics_masked(state, state_mask, action) — left the
    1e-4 gate.  Walk the REFERENCE trajectory (oracle trace) and at every step run the engine for ONE step from the reference
    state (identical graph): its prediction must stay inside the gate, and the first step at which the engine's rebuilt edge
    list differs from the reference's must differ only by candidates whose distance gap (to the competing candidate at the
    top-k cut, or to the radius) is < 10x that step's measured position deviation.  Returns (step, deviation, gap); raises
    AssertionError if the divergence has any other cause.  `allow_drift` (long rollouts): identical edge lists at every step with
    every one-step deviation inside the gate is accepted too — the final difference is then the accumulated per-step rounding
    ("gate parity per step on identical graphs; report rollout drift separately") — returns (None, max step deviation, 0)."""
    task = configs.task_config(material)
    mm = synth.MATERIALS[material]
    trace = []
    if state_mask is None:
        ago.dynamics(weights, task, state, action[b:b + 1], trace=trace)
        height_mode, obj_mask = _lib.AG_HEIGHT_MIN, None
    else:
        ago.dynamics_masked(weights, task, state[b:b + 1], state_mask[b:b + 1], action[b:b + 1], trace=trace)
        height_mode, obj_mask = _lib.AG_HEIGHT_MASKED_MEAN, t(np.asarray(state_mask[b:b + 1], bool))
    raise_by = 0.01 * task["sim_real_ratio"] if task["gripper_enable"] else 0.0
    assert len(trace) >= 2, "a one-step rollout cannot diverge through an edge flip"
    one = torch.ones(1, dtype=torch.int32, device=DEV)
    max_dev = 0.0
    for ai in range(len(trace) - 1):
        tr, nx = trace[ai], trace[ai + 1]
        thr = aggraph.threshold_sq(tr["radius"], 1, torch.device(DEV), _lib.AG_VARIANT_BATCH)
        _, fin = rollout(model, t(tr["states"]), t(tr["delta"]), t(tr["attrs"]), t(tr["p_instance"]), t(tr["phys"]), t(tr["mask"]),
                         t(tr["tool_mask"]), thr, one, 1, mm["topk"], mm["connect_tools_all"], mm["n_tools"], height_mode, obj_mask,
                         raise_by, return_state=True)
        cur_e, cur_r = fin[0, -1].cpu().numpy(), nx["states"][0, -1]
        dev = float(np.abs(cur_e - cur_r).max())
        max_dev = max(max_dev, dev)
        assert dev <= TOL_FWD, f"step {ai + 1}: one-step deviation {dev} on the identical graph"
        csr = aggraph.build_edges(fin[:, -1].contiguous(), tr["radius"], t(tr["mask"]), t(tr["tool_mask"]), mm["topk"],
                                  mm["connect_tools_all"], "batch", max_tools=mm["n_tools"])
        er, es = csr.to_lists()[0]
        n = int(nx["n_rel"][0])
        ref = set(zip(nx["recv"][0, :n].tolist(), nx["send"][0, :n].tolist()))
        got = set(zip(er.tolist(), es.tolist()))
        if ref == got:
            continue
        radius = float(np.asarray(tr["radius"]).reshape(-1)[0])
        dist = lambda i, j: float(np.linalg.norm(cur_r[i].astype(np.float64) - cur_r[j].astype(np.float64)))
        worst = 0.0
        for i in {e[0] for e in ref ^ got}:
            only_ref = [j for (r, j) in ref - got if r == i]
            only_got = [j for (r, j) in got - ref if r == i]
            d_ref, d_got = sorted(dist(i, j) for j in only_ref), sorted(dist(i, j) for j in only_got)
            for k in range(max(len(d_ref), len(d_got))):       # swapped candidates pair up; an unpaired one sits at the radius
                a = d_ref[k] if k < len(d_ref) else radius
                c = d_got[k] if k < len(d_got) else radius
                worst = max(worst, abs(a - c))
        assert worst < 10 * max(dev, 1e-7), f"step {ai + 1}: edge lists differ by candidates {worst} apart, deviation {dev}"
        return ai + 1, dev, worst
    if allow_drift:
        return None, max_dev, 0.0
    raise AssertionError("final states differ but every rebuilt edge list equals the reference's")


@pytest.mark.parametrize("name", golden_files("dyn_"))
def test_dynamics_golden(name, weights, prec):
    g = load_golden(name)
    material = str(g["material"])
    weights = weights_for(g, weights)
    m = make_model(weights, material, prec=prec)
    out = dynamics(t(g["state"]), t(g["action"]), m, DEV, _ppm(material))
    assert m.take_status() == 0
    assert out["state_seqs"].shape == g["state_seqs"].shape
    assert np.abs(out["action_seqs"].cpu().numpy() - g["action_seqs"]).max() <= 1e-6
    err = np.abs(out["state_seqs"].cpu().numpy() - g["state_seqs"]).reshape(g["state_seqs"].shape[0], -1).max(1)
    assert prec != "f32" or (err <= TOL_FWD).all(), f"exact-fp32 mode must match every reference rollout: {err}"
    for b in np.nonzero(err > TOL_FWD)[0]:
        step, dev, gap = explain_divergence(m, weights, material, g["state"], g["action"], int(b))
        print(f"{name}[{b}] ({prec}): top-k near-tie at step {step}: candidates {gap:.2e} apart, forward deviation {dev:.2e}")


@pytest.mark.parametrize("material,n_obj,kw", [("granular", 2000, {}), ("cloth", 4096, {})])
def test_rollout_full_shape_vs_oracle(material, n_obj, kw, weights, prec):
    """BASELINE configs[2] / [3] graph shapes under a ROLLOUT (edge rebuild from predicted positions, tool advance,
    connect_tools_all for cloth with the tool near the sheet in sample 0 and far from it in sample 1): 3 model steps,
    batch 2, engine vs the oracle; exact-fp32 mode inside the gate outright, split-bf16 modes inside it or a proven
    near-tie."""
    state, act = synth.make_mpc_inputs(material, n_obj, 2, seed=21, len_lo=3, len_hi=3.9, **kw)
    c = state.mean(0)
    act[0, 0, 0], act[0, 0, 1] = c[0], c[2]              # push starts over the middle of the cloud: tool edges exist
    act[1, 0, 0], act[1, 0, 1] = c[0] + 500.0, c[2]      # far away: no tool-object edge, batch_mask stays false (graph.py:123,135)
    ref, _ = ago.dynamics(weights, configs.task_config(material), state, act)
    m = make_model(weights, material, prec=prec)
    out = dynamics(t(state), t(act), m, DEV, _ppm(material))["state_seqs"].cpu().numpy()
    err = np.abs(out - ref).reshape(2, -1).max(1)
    assert float(np.abs(ref[0, 0] - state).max()) > 1e-3, "the pushed cloud must actually move"
    assert prec != "f32" or (err <= TOL_FWD).all(), err
    for b in np.nonzero(err > TOL_FWD)[0]:
        print(material, prec, explain_divergence(m, weights, material, state, act, int(b)))


@pytest.mark.parametrize("material,n_obj,batch,steps", [("cloth", 4096, 64, 20), ("rope", 1000, 256, 10), ("granular", 2000, 128, 10)])
def test_rollout_at_the_benchmarked_config_vs_oracle(material, n_obj, batch, steps, weights, prec):
    """BASELINE configs[3]'s per-GPU share (cloth-4k, batch 64, 20-step rollout), configs[1] (rope-1k, batch 256, 10 steps) and configs[2]
    (granular-2k, batch 128, 10 steps as bench.py times it) at FULL size:
    four samples of the batch against the oracle's rollout of the same samples (exact-fp32 mode inside the gate outright; the split
    modes inside it or a proven top-k near-tie), and those samples rolled out alone equal their rows of the full-batch result bit for
    bit (batch-composition independence at the benchmarked shape, two rollout streams included)."""
    kw = dict(spacing=0.1) if material == "rope" else {}
    state, act = synth.make_mpc_inputs(material, n_obj, batch, seed=33, len_lo=steps, len_hi=steps + 0.9, **kw)
    c = state.mean(0)
    pick = [0, batch // 3, 2 * batch // 3, batch - 1]
    for k, b in enumerate(pick):                       # pushes that start over the cloud: tool edges exist from the first step
        act[b, 0, 0], act[b, 0, 1] = state[(k * 997) % n_obj, 0], state[(k * 997) % n_obj, 2]
    m = make_model(weights, material, prec=prec)
    full = dynamics(t(state), t(act), m, DEV, _ppm(material))["state_seqs"]
    assert m.take_status() == 0
    sub = dynamics(t(state), t(act[pick]), m, DEV, _ppm(material))["state_seqs"]
    assert torch.equal(full[pick], sub), "a sample's rollout must not depend on the batch it is in"
    ref, _ = ago.dynamics(weights, configs.task_config(material), state, act[pick])
    err = np.abs(sub.cpu().numpy() - ref).reshape(len(pick), -1).max(1)
    assert float(np.abs(ref[0, 0] - state).max()) > 1e-3, "the pushed cloud must actually move"
    # exact-fp32 mode: inside the gate after all steps, or — granular-2k saturates its top-20 with candidates a few 1e-7 apart — a PROVEN
    # top-k near-tie (explain_divergence raises unless the first differing edge list differs only by such candidates); no drift allowance
    assert prec != "f32" or material == "granular" or (err <= TOL_FWD).all(), err
    record = os.environ.get("AG_DRIFT_FILE")    # evidence runs (tools/evidence_r05.sh): the measured drift is RECORDED, one line per sample
    for b in (range(len(pick)) if record else np.nonzero(err > TOL_FWD)[0]):      # 10-20 steps: per-step parity on identical graphs is the gate
        if err[b] > TOL_FWD:
            step, dev, gap = explain_divergence(m, weights, material, state, act[pick], int(b), allow_drift=(prec != "f32"))
        else:       # inside the gate after all steps: nothing to prove; the one-step deviations are measured for the record only
            try:
                step, dev, gap = explain_divergence(m, weights, material, state, act[pick], int(b), allow_drift=True)
            except AssertionError as e:      # (a proven near-tie that healed, or a tie explain_divergence cannot classify: not a failure inside the gate)
                step, dev, gap = None, float("nan"), 0.0
                print("note:", e)
        line = (f"{material} {n_obj} x {batch} x {steps} steps | {prec:6s} | sample {pick[b]:3d} | final-state drift vs the reference rollout {err[b]:.2e}"
                f" | max one-step deviation on identical graphs {dev:.2e} | "
                + (f"top-k near-tie at step {step} (candidates {gap:.2e} apart)" if step else "edge lists equal the reference's at every step"))
        print(line)
        if record:
            with open(record, "a") as f:
                f.write(line + "\n")


def test_rollout_at_the_global_batch_of_configs3(weights):
    """BASELINE configs[3] at its GLOBAL batch on one GPU: cloth-4k, batch 512, 20-step rollout (8 GPUs would run 64 each).  The first and the
    last per-GPU share (64 samples each) rolled out alone equal their rows of the 512-batch result bit for bit, and two samples are held
    against the oracle's rollout like the per-GPU share in test_rollout_at_the_benchmarked_config_vs_oracle."""
    material, n_obj, batch, steps = "cloth", 4096, 512, 20
    state, act = synth.make_mpc_inputs(material, n_obj, batch, seed=34, len_lo=steps, len_hi=steps + 0.9)
    pick = [0, batch - 1]
    for k, b in enumerate(pick):
        act[b, 0, 0], act[b, 0, 1] = state[(k * 997) % n_obj, 0], state[(k * 997) % n_obj, 2]
    m = make_model(weights, material, prec="fast")
    full = dynamics(t(state), t(act), m, DEV, _ppm(material))["state_seqs"]
    assert m.take_status() == 0 and full.shape == (batch, 1, n_obj, 3) and torch.isfinite(full).all()
    for lo in (0, batch - 64):
        share = dynamics(t(state), t(act[lo:lo + 64]), m, DEV, _ppm(material))["state_seqs"]
        assert torch.equal(full[lo:lo + 64], share), "a per-GPU share must equal its rows of the global batch"
    ref, _ = ago.dynamics(weights, configs.task_config(material), state, act[pick])
    err = np.abs(full[pick].cpu().numpy() - ref).reshape(len(pick), -1).max(1)
    for b in np.nonzero(err > TOL_FWD)[0]:
        step, dev, gap = explain_divergence(m, weights, material, state, act[pick], int(b), allow_drift=True)
        print(f"cloth 512 x 20 sample {pick[b]}: drift {err[b]:.2e}, max one-step deviation {dev:.2e}, near-tie step {step}")


@pytest.mark.parametrize("bad", ["nan", "-nan", "inf", "-inf"])
def test_nonfinite_inputs_raise_the_status_in_both_edge_kernels(weights, bad):
    """The weight-stationary and the streaming edge encoder are bit-identical AND report alike: a non-finite position in a particle that
    has edges raises status bit 0 with either kernel (the weight-stationary kernel's table maximum dropped NaNs and built its
    first-layer operands without a range check, so a NaN could end as a stored 0 with status 0)."""
    g = synth.make_graph_inputs("rope", 200, 2, seed=9, spacing=0.1)
    csr = aggraph.build_edges(t(g["state"][:, -1]), 0.5, t(g["mask"]), t(g["tool_mask"]), 10, False, "batch", max_tools=1)   # graph from the clean cloud
    val = {"nan": np.float32(np.nan), "-nan": -np.float32(np.nan), "inf": np.float32(np.inf), "-inf": -np.float32(np.inf)}[bad]
    st = g["state"].copy()
    st[1, -1, 57, 1] = val                     # current frame of one particle of sample 1
    m = make_model(weights, prec="fast")
    kw = dict(action=t(g["action"]), rope_physics_param=t(g["phys"]))
    flags = []
    for ws in (0, 1):
        m.set_option("edge_stationary", ws)
        m.take_status()
        import warnings as w
        with w.catch_warnings():
            w.simplefilter("ignore")
            _, mot = m(t(st), t(g["attrs"]), csr, None, t(g["p_instance"]), **kw)
            flags.append(m.take_status() & 1)
        assert torch.isfinite(mot[0]).all()    # the other sample is untouched
    assert flags == [1, 1], flags
    m.set_option("edge_stationary", 1)
    _, clean = m(t(g["state"]), t(g["attrs"]), csr, None, t(g["p_instance"]), **kw)
    assert m.take_status() == 0 and torch.isfinite(clean).all()


def test_strict_status_raises_in_the_call_that_caused_it(weights, monkeypatch):
    """AG_STRICT_STATUS=1: dynamics() reads the numeric status at its END and raises (default: the next call's read warns — sync-free)."""
    big = {k: v.copy() for k, v in weights.items()}
    for k in ("relation_encoder.model.0.weight", "relation_encoder.model.0.bias", "relation_encoder.model.2.weight"):
        big[k] *= 1.0e3                                          # a hidden fp16 activation beyond 65504 (see the overflow test below)
    state, act = synth.make_mpc_inputs("rope", 100, 4, seed=1, len_lo=2, len_hi=3.9, spacing=0.1)
    act[:, 0, 0], act[:, 0, 1] = state[50, 0], state[50, 2]
    m = make_model(big, prec="fast")
    monkeypatch.setenv("AG_STRICT_STATUS", "1")
    with pytest.raises(FloatingPointError):
        dynamics(t(state), t(act), m, DEV, _ppm("rope"))
    assert m.take_status() == 0                                  # consumed by the raising call
    ok = make_model(weights, prec="fast")
    out = dynamics(t(state), t(act), ok, DEV, _ppm("rope"))["state_seqs"]
    assert torch.isfinite(out).all()
    monkeypatch.delenv("AG_STRICT_STATUS")
    import warnings as w
    with w.catch_warnings(record=True) as rec:                   # default: deferred by one call, as a warning
        w.simplefilter("always")
        dynamics(t(state), t(act), m, DEV, _ppm("rope"))
        assert not any("non-finite" in str(x.message) for x in rec)
        dynamics(t(state), t(act), m, DEV, _ppm("rope"))
        assert any("non-finite" in str(x.message) for x in rec)


@pytest.mark.parametrize("n_obj", [100, 300])
def test_nonfinite_state_raises_the_status_through_the_rollouts_riders(weights, n_obj):
    """In ag_rollout the per-node input rows of the edge features are written by rider workgroups of the edge builder's bin_kernel (uniform-grid
    path, N >= 256) instead of edge_node_tab_kernel (brute-force path: still that kernel): the check of the raw inputs travels with them, so a
    non-finite position raises status bit 0 on either path, and a clean call raises nothing."""
    import warnings as w
    state, act = synth.make_mpc_inputs("rope", n_obj, 4, seed=3, len_lo=2, len_hi=3.9, spacing=0.1)
    m = make_model(weights, prec="fast")
    m.take_status()
    out = dynamics(t(state), t(act), m, DEV, _ppm("rope"))["state_seqs"]
    assert m.take_status() == 0 and torch.isfinite(out).all()
    bad = state.copy()
    bad[n_obj // 2, 1] = np.float32(np.nan)
    with w.catch_warnings():
        w.simplefilter("ignore")
        dynamics(t(bad), t(act), m, DEV, _ppm("rope"))
        assert m.take_status() & 1


@pytest.mark.parametrize("split", [32, 96, 224])
def test_cu_partitioned_rollout_is_bitwise_the_shared_chip_rollout(weights, split):
    """ag_set_option("cu_split", X): the edge encoder on X / 8 CUs of every XCD, everything else on the other CUs, two CU-masked queues with
    the batch parts pipelined through them (measured slower than sharing the chip, off by default).  Same kernels, same
    per-row arithmetic: bit-identical results, repeatable, and the option switches back cleanly."""
    state, act = synth.make_mpc_inputs("rope", 300, 24, seed=6, len_lo=3, len_hi=5.9, spacing=0.1)
    m = make_model(weights, "rope", prec="fast")
    ref = dynamics(t(state), t(act), m, DEV, _ppm("rope"))["state_seqs"]
    m.set_option("cu_split", split)
    for _ in range(3):
        assert torch.equal(dynamics(t(state), t(act), m, DEV, _ppm("rope"))["state_seqs"], ref)
    m.set_option("rollout_streams", 3)
    assert torch.equal(dynamics(t(state), t(act), m, DEV, _ppm("rope"))["state_seqs"], ref)
    m.set_option("cu_split", 0)
    assert torch.equal(dynamics(t(state), t(act), m, DEV, _ppm("rope"))["state_seqs"], ref) and m.take_status() == 0
    with pytest.raises(RuntimeError):
        m.set_option("cu_split", 100)           # not a multiple of 8


def test_get_option_reports_what_the_engine_runs_with(weights):
    """ag_get_option: the model's current value of every option of ag_set_option (defaults, then what was set); unknown names are an error code."""
    import ctypes
    m = make_model(weights, "rope", prec="fast")
    L, h = _lib.lib(), m.handle(torch.device(DEV))

    def get(name):
        v = ctypes.c_int(-99)
        _lib.check(L.ag_get_option(h, name.encode(), ctypes.byref(v)), name)
        return v.value

    assert (get("precision"), get("self_edges"), get("shared_state"), get("node_dedup"), get("rollout_streams"), get("fuse_aggregate"), get("agg_q16")) == (2, 1, 0, 1, 0, 0, 1)
    for name, val in (("precision", 1), ("precision", 0), ("self_edges", 0), ("shared_state", 1), ("node_dedup", 2), ("rollout_streams", 3), ("edge_products", 3),
                      ("edge_stationary", 0), ("node_stationary", 0), ("cu_split", 64), ("fuse_aggregate", 2), ("agg_q16", 0)):
        m.set_option(name, val)
        assert get(name) == val and m.get_option(name) == val, name
    v = ctypes.c_int()
    assert L.ag_get_option(h, b"no_such_option", ctypes.byref(v)) != 0 and b"unknown option" in L.ag_last_error()


def test_rollout_stream_count_default_follows_the_workload_and_never_changes_a_bit(weights):
    """"rollout_streams" 0 (the default): one stream, two where the edge stack dominates the step (top-k >= 16: granular), never more than B / 8
    parts; 1..4 are taken as given (ag_rollout_streams_for reports what ag_rollout will do).  Graphs never interact: every count gives the same bits."""
    import ctypes
    m = make_model(weights, "rope", prec="fast")
    L, h = _lib.lib(), m.handle(torch.device(DEV))

    def parts(B, topk):
        prm = _lib.RolloutParams(B, 301, 300, 1, topk, 0, 1, 4, 0, 0.0)
        return L.ag_rollout_streams_for(h, ctypes.byref(prm))

    assert (parts(256, 10), parts(256, 5), parts(256, 20), parts(12, 20), parts(4, 20)) == (1, 1, 2, 1, 1)
    state, act = synth.make_mpc_inputs("rope", 300, 40, seed=9, len_lo=3, len_hi=5.9, spacing=0.1)
    ref = dynamics(t(state), t(act), m, DEV, _ppm("rope"))["state_seqs"]
    for n in (1, 2, 3, 4):
        m.set_option("rollout_streams", n)
        assert parts(256, 10) == n and parts(20, 10) == min(n, 2)
        assert torch.equal(dynamics(t(state), t(act), m, DEV, _ppm("rope"))["state_seqs"], ref)
    m.set_option("rollout_streams", 0)
    assert parts(256, 10) == 1
    with pytest.raises(RuntimeError):
        m.set_option("rollout_streams", 5)


def test_rollout_is_hip_graph_capturable(weights):
    """ag_rollout never synchronises the host and joins its auxiliary streams on every path, so a caller may capture it in a HIP graph
    (torch.cuda.graphs): the replayed rollout equals the enqueued one bit for bit, twice (replay is not faster, the point is
    that nothing in the library prevents it)."""
    from adaptigraph_amd.forward_dynamics import rollout as ag_roll
    m = make_model(weights, prec="fast")
    B, T = 24, 4
    g = synth.make_graph_inputs("rope", 200, B, seed=3, spacing=0.1)
    dev = torch.device(DEV)
    thr = aggraph.threshold_sq(0.5, B, dev, _lib.AG_VARIANT_BATCH)
    rep = torch.tensor([(b % T) + 1 for b in range(B)], dtype=torch.int32, device=dev)
    args = (m, t(g["state"]), t(g["action"]), t(g["attrs"]), t(g["p_instance"]), t(g["phys"]), t(g["mask"]), t(g["tool_mask"]), thr, rep, T, 10, False, 1)
    ref = ag_roll(*args).clone()
    if os.environ.get("AG_SHARED_STATE") == "1":      # (suite-wide run of the shared-state rollout: the eager result must be the plain path's)
        m.set_option("shared_state", 0)
        assert torch.equal(ref, ag_roll(*args)), "eager shared-state rollout differs from the plain one"
        m.set_option("shared_state", 1)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ag_roll(*args)
    torch.cuda.current_stream().wait_stream(side)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        out = ag_roll(*args)
    for _ in range(3):
        out.zero_()
        gr.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ref)
    # ... and with what a planner would switch on: the shared-state rollout and a forced node de-duplication (device words zeroed per call — by a
    # kernel: r06 found hipMemsetAsync NODES of a captured graph not replayed reliably, the second replay ran with the first one's counters)
    for name, val in (("node_dedup", 2), ("shared_state", 1)):
        m.set_option(name, val)
        with torch.cuda.stream(side):
            assert torch.equal(ag_roll(*args), ref)
        torch.cuda.current_stream().wait_stream(side)
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2):
            out2 = ag_roll(*args)
        for _ in range(3):
            out2.zero_()
            g2.replay()
            torch.cuda.synchronize()
            assert torch.equal(out2, ref), name
    assert m.take_status() == 0


@pytest.mark.parametrize("name", golden_files("dynmask_"))
def test_dynamics_masked_golden(name, weights, prec):
    g = load_golden(name)
    material = str(g["material"])
    m = make_model(weights, material, prec=prec)
    out = dynamics_masked(t(g["state_init"]), t(g["state_mask"]), t(g["action"]), m, DEV, _ppm(material))
    assert np.abs(out["action_seqs"].cpu().numpy() - g["action_seqs"]).max() <= 1e-6
    err = np.abs(out["state_seqs"].cpu().numpy() - g["state_seqs"]).reshape(g["state_seqs"].shape[0], -1).max(1)
    assert prec != "f32" or (err <= TOL_FWD).all(), f"exact-fp32 mode must match every reference rollout: {err}"
    for b in np.nonzero(err > TOL_FWD)[0]:      # inside the gate, or a PROVEN top-k near-tie (see explain_divergence)
        step, dev, gap = explain_divergence(m, weights, material, g["state_init"], g["action"], int(b), state_mask=g["state_mask"])
        print(f"{name}[{b}] ({prec}): top-k near-tie at step {step}: candidates {gap:.2e} apart, forward deviation {dev:.2e}")


def test_dynamics_vs_oracle_rope300(weights, prec):
    state, act = synth.make_mpc_inputs("rope", 300, 5, seed=12, len_lo=2, len_hi=5, spacing=0.1)
    seq_ref, dec_ref = ago.dynamics(weights, configs.task_config("rope"), state, act)
    m = make_model(weights, "rope", prec=prec)
    out = dynamics(t(state), t(act), m, DEV, _ppm("rope"))
    assert np.abs(out["state_seqs"].cpu().numpy() - seq_ref).max() <= TOL_FWD


def test_rollout_full_size_matches_small_batch(weights, model):
    """BASELINE configs[1] shape (rope ~1k particles, batch 256, 10 steps): sample b of the big batch must be
    bit-identical to the same sample rolled out alone (size-independent property; the oracle would need minutes)."""
    B, T = 256, 10
    state, act = synth.make_mpc_inputs("rope", 1000, B, seed=3, len_lo=T, len_hi=T + 0.9, spacing=0.1)
    ppm = _ppm("rope")
    out = dynamics(t(state), t(act), model, DEV, ppm)["state_seqs"]
    assert out.shape == (B, 1, 1000, 3) and torch.isfinite(out).all()
    for b in (0, 131, 255):
        one = dynamics(t(state), t(act[b:b + 1]), model, DEV, ppm)["state_seqs"]
        assert torch.equal(out[b], one[0])
    assert (out[:, 0] - t(state)[None]).abs().max().item() > 1e-3      # the rope actually moved


@pytest.mark.parametrize("material,n_obj", [("granular", 200), ("rope", 300), ("cloth", 256)])
def test_rollout_per_step_on_identical_graphs(weights, material, n_obj):
    """Teacher-forced rollout: at every step both arithmetic modes start from the SAME state (the exact-fp32
    trajectory), hence identical graphs; the split-bf16 step must stay within the 1e-4 gate of the exact step."""
    g = synth.make_graph_inputs(material, n_obj, 4, seed=6)
    mm = synth.MATERIALS[material]
    exact, fast = make_model(weights, material, prec="f32"), make_model(weights, material, prec="fast")
    thr = aggraph.threshold_sq(mm["radius"], 4, torch.device(DEV), _lib.AG_VARIANT_BATCH)
    one = torch.ones(4, dtype=torch.int32, device=DEV)
    state = t(g["state"])
    args = (t(g["action"]), t(g["attrs"]), t(g["p_instance"]), t(g["phys"]), t(g["mask"]), t(g["tool_mask"]), thr, one, 1,
            mm["topk"], mm["connect_tools_all"], g["n_tools"])
    worst = 0.0
    for _ in range(5):
        ref, nxt = rollout(exact, state, *args, return_state=True)
        got = rollout(fast, state, *args)
        worst = max(worst, (ref - got).abs().max().item())
        state = nxt
    assert 0 < worst <= TOL_FWD, worst


def test_rollout_zero_steps_and_unreached_repeat(weights, model):
    g = synth.make_graph_inputs("rope", 50, 3, seed=5, spacing=0.1)
    thr = aggraph.threshold_sq(0.5, 3, torch.device(DEV), _lib.AG_VARIANT_BATCH)
    rep = torch.tensor([2, 0, 7], dtype=torch.int32, device=DEV)       # 0: never recorded; 7 > n_steps: never reached
    seq, fin = rollout(model, t(g["state"]), t(g["action"]), t(g["attrs"]), t(g["p_instance"]), t(g["phys"]),
                       t(g["mask"]), t(g["tool_mask"]), thr, rep, 3, 10, False, 1, return_state=True)
    assert seq[1].abs().sum() == 0 and seq[2].abs().sum() == 0 and seq[0].abs().sum() > 0
    # history shift: after 3 steps with n_his = 4 the oldest kept frame is the original newest frame
    assert torch.equal(fin[:, 0], t(g["state"])[:, 3]) and torch.isfinite(fin).all()
    seq0 = rollout(model, t(g["state"]), t(g["action"]), t(g["attrs"]), t(g["p_instance"]), t(g["phys"]),
                   t(g["mask"]), t(g["tool_mask"]), thr, rep, 0, 10, False, 1)
    assert seq0.abs().sum() == 0


def test_mppi_iteration_full_size_equals_chunked(weights):
    """BASELINE configs[4] shape: one MPPI iteration with 1024 sampled pushes x 15 model steps on rope-1k.  Size-independent
    property: rolling the samples out as one batch or in 4 chunks of 256 gives bit-identical predicted states (rows never
    interact; chunk boundaries move samples between row tiles and rollout streams), hence identical rewards and update."""
    from functools import partial
    from adaptigraph_amd import losses, mpc
    task = configs.task_config("rope")
    lo, hi = np.array(task["action_lower_lim"], np.float32), np.array(task["action_upper_lim"], np.float32)
    lo[3], hi[3] = 15, 15.5
    state, act = synth.make_mpc_inputs("rope", 1000, 1, seed=0, len_lo=15, len_hi=15.4, spacing=0.1)
    target = (state + np.array([0.4, 0.0, 0.3], np.float32)).astype(np.float32)
    bbox = np.array([[state[:, 0].min() - 5, state[:, 0].max() + 5], [state[:, 2].min() - 5, state[:, 2].max() + 5]])
    m = make_model(weights, "rope", prec="fast")
    planner = mpc.MPPIPlanner(m, DEV, _ppm("rope"), partial(losses.chamfer, y=t(target)[None]),
                              partial(losses.rope_penalty, sim_real_ratio=task["sim_real_ratio"]), bbox, lo, hi,
                              n_sample=1024, n_update_iter=1, rollout_best=False)
    torch.manual_seed(7)
    samples = planner.sample(t(act[0]), 1)
    assert samples.shape == (1024, 1, 4)
    new_seq, reward, out = planner.step(t(state), samples)
    seqs = out["state_seqs"]
    assert seqs.shape == (1024, 1, 1000, 3) and torch.isfinite(seqs).all() and torch.isfinite(reward).all()
    chunks = torch.cat([dynamics(t(state), samples[i:i + 256], m, DEV, _ppm("rope"))["state_seqs"] for i in range(0, 1024, 256)])
    assert torch.equal(seqs, chunks)
    r2 = planner.evaluate_traj(chunks, samples, state_cur=t(state))["reward_seqs"]
    assert torch.equal(reward, r2) and new_seq.shape == (1, 4)
    assert float(reward.max() - reward.min()) > 0


def test_undersized_workspace_is_an_error_not_a_fault(weights, model):
    """AG_ERR_WS: the library never touches a workspace smaller than *_workspace_bytes says it needs."""
    import ctypes
    g = synth.make_graph_inputs("rope", 100, 2, seed=1, spacing=0.1)
    csr = aggraph.build_edges(t(g["state"][:, -1]), 0.5, t(g["mask"]), t(g["tool_mask"]), 10, False, "batch", max_tools=1)
    L = _lib.lib()
    need = L.ag_forward_workspace_bytes(2, 101, csr.e_cap)
    hm = model.handle(torch.device(DEV))
    assert L.ag_forward_workspace_bytes_for(hm, 2, 101, csr.e_cap) <= need
    ws = torch.empty(need // 2, dtype=torch.uint8, device=DEV)
    pos, mot = torch.full((2, 100, 3), 7.0, device=DEV), torch.full((2, 100, 3), 7.0, device=DEV)
    st, at, ac, pi, ph = (t(g[k]) for k in ("state", "attrs", "action", "p_instance", "phys"))
    rc = L.ag_forward(model.handle(torch.device(DEV)), st.data_ptr(), at.data_ptr(), ac.data_ptr(), pi.data_ptr(), 1, ph.data_ptr(),
                      csr.row_ptr.data_ptr(), csr.edge_recv.data_ptr(), csr.edge_send.data_ptr(), csr.e_cap, 2, 101, 100,
                      pos.data_ptr(), mot.data_ptr(), ws.data_ptr(), ws.numel(), None)
    assert rc == -3 and b"workspace" in L.ag_last_error()
    torch.cuda.synchronize()
    assert bool((pos == 7.0).all()) and bool((mot == 7.0).all())       # nothing was launched
    prm = _lib.RolloutParams(2, 101, 100, 1, 10, 0, 1, 2, 0, 0.0)
    need_r = L.ag_rollout_workspace_bytes(ctypes.byref(prm))
    u8 = lambda x: t(x).view(torch.uint8)
    thr = aggraph.threshold_sq(0.5, 2, torch.device(DEV), _lib.AG_VARIANT_BATCH)
    rep = torch.ones(2, dtype=torch.int32, device=DEV)
    out = torch.zeros((2, 100, 3), device=DEV)
    rc = L.ag_rollout(model.handle(torch.device(DEV)), ctypes.byref(prm), st.data_ptr(), ac.data_ptr(), at.data_ptr(), pi.data_ptr(),
                      ph.data_ptr(), u8(g["mask"]).data_ptr(), u8(g["tool_mask"]).data_ptr(), None, thr.data_ptr(), rep.data_ptr(),
                      out.data_ptr(), None, ws.data_ptr(), min(ws.numel(), need_r // 4), None)
    assert rc == -3
    # and the model is still usable afterwards (no state was left half-changed)
    _, again = model(st, at, csr, None, pi, action=ac, rope_physics_param=ph)
    assert torch.isfinite(again).all()


def test_workspace_is_bounded_and_sized_for_the_models_mode(weights):
    """ag_*_workspace_bytes_for(model, ...): exact for the model's current mode — the default mode's per-edge table is 16-bit rows, half the
    largest buffer — and the compact tables of the node de-duplication are bounded.  Targets of VERDICT r04 item 3: rope-1k x 256 x 10 <= 2.9 GB,
    the reference planner's 20 000 x 200 <= 40 GB.  A workspace sized for the default mode is refused (AG_ERR_WS) once the model needs fp32 rows."""
    import ctypes
    L = _lib.lib()
    m = make_model(weights, prec="fast")
    h = m.handle(torch.device(DEV))
    c2 = _lib.RolloutParams(256, 1001, 1000, 1, 10, 0, 1, 10, 0, 0.0)
    plan = _lib.RolloutParams(20000, 201, 200, 1, 10, 0, 1, 15, 0, 0.0)
    any_c2, any_plan = L.ag_rollout_workspace_bytes(ctypes.byref(c2)), L.ag_rollout_workspace_bytes(ctypes.byref(plan))
    fast_c2, fast_plan = L.ag_rollout_workspace_bytes_for(h, ctypes.byref(c2)), L.ag_rollout_workspace_bytes_for(h, ctypes.byref(plan))
    print(f"rollout workspace: rope-1k x 256: {fast_c2 / 1e9:.2f} GB (any mode {any_c2 / 1e9:.2f}); 20 000 x 200: {fast_plan / 1e9:.1f} GB (any mode {any_plan / 1e9:.1f})")
    assert fast_c2 <= 2.2e9 and fast_c2 <= any_c2 <= 2.95e9
    assert fast_plan <= 40e9 and fast_plan <= any_plan <= 47e9
    m.set_option("precision", 0)
    assert L.ag_rollout_workspace_bytes_for(h, ctypes.byref(c2)) == any_c2
    # sized for the 16-bit table, called in a mode that needs fp32 rows: an error code, nothing launched
    g = synth.make_graph_inputs("rope", 100, 2, seed=1, spacing=0.1)
    csr = aggraph.build_edges(t(g["state"][:, -1]), 0.5, t(g["mask"]), t(g["tool_mask"]), 10, False, "batch", max_tools=1)
    m.set_option("precision", 2)
    small = L.ag_forward_workspace_bytes_for(h, 2, 101, csr.e_cap)
    ws = torch.empty(small, dtype=torch.uint8, device=DEV)
    pos, mot = torch.full((2, 100, 3), 7.0, device=DEV), torch.full((2, 100, 3), 7.0, device=DEV)
    st, at, ac, pi, ph = (t(g[k]) for k in ("state", "attrs", "action", "p_instance", "phys"))
    call = lambda: L.ag_forward(h, st.data_ptr(), at.data_ptr(), ac.data_ptr(), pi.data_ptr(), 1, ph.data_ptr(), csr.row_ptr.data_ptr(),
                                csr.edge_recv.data_ptr(), csr.edge_send.data_ptr(), csr.e_cap, 2, 101, 100, pos.data_ptr(), mot.data_ptr(),
                                ws.data_ptr(), ws.numel(), None)
    assert call() == 0
    torch.cuda.synchronize()
    assert torch.isfinite(mot).all() and not bool((mot == 7.0).all())
    m.set_option("precision", 1)
    assert call() == -3 and b"workspace" in L.ag_last_error()


@pytest.mark.parametrize("n_obj", [300, 1000])
def test_edge_builder_terminates_on_nonfinite_positions(n_obj):
    """A diverged rollout can hand the uniform-grid builder (N >= 256) NaN or Inf coordinates.  The reference yields no
    edge for a non-finite distance (graph.py:121 `dis < thresh` is false); the grid set-up must not spin on non-finite
    extents (r01: the cell-size doubling loop never exited) and the edge lists must still equal the oracle's."""
    g = synth.make_graph_inputs("rope", n_obj, 3, seed=13, spacing=0.1)
    pos = g["state"][:, -1].copy()
    pos[0, 5] = np.nan                      # one NaN particle
    pos[1, 7, 0] = np.inf                   # one Inf coordinate: non-finite extent
    pos[1, 9] = -np.inf
    pos[2, :, 1] = np.nan                   # a whole sample without finite distances
    csr = aggraph.build_edges(t(pos), 0.5, t(g["mask"]), t(g["tool_mask"]), 10, False, "batch", max_tools=1)
    torch.cuda.synchronize()
    n_rel, recv, send = ago.build_edges(pos, 0.5, g["mask"], g["tool_mask"], 10, False, "batch")
    got = csr.to_lists()
    for b in range(3):
        assert np.array_equal(got[b][0], recv[b, :n_rel[b]]) and np.array_equal(got[b][1], send[b, :n_rel[b]]), b
    assert n_rel[2] == 0 and not (got[0][0] == 5).any() and not (got[0][1] == 5).any()


def test_fp16_activation_overflow_is_reported_where_it_happens(weights):
    """Precision mode 2 runs the edge stack on fp16 activations.  With the relation encoder scaled so that a HIDDEN activation exceeds
    65504 the sticky status bit is raised by the epilogue that produced the inf (and warned about at the next rollout call) — whatever the
    following layers make of it (inf x negative weight -> -inf -> ReLU -> 0 would otherwise hide it); mode 1 on the same weights stays
    finite and raises nothing.  The per-edge table itself (block-scaled q16) has no range limit: |Eterm| ~ 1e8 is stored and summed."""
    import warnings as w
    g = synth.make_graph_inputs("rope", 100, 2, seed=3, spacing=0.1)
    csr = aggraph.build_edges(t(g["state"][:, -1]), 0.5, t(g["mask"]), t(g["tool_mask"]), 10, False, "batch", max_tools=1)
    args = lambda: (t(g["state"]), t(g["attrs"]), csr, None, t(g["p_instance"]))
    kw = dict(action=t(g["action"]), rope_physics_param=t(g["phys"]))
    big = {k: v.copy() for k, v in weights.items()}
    for k in ("relation_encoder.model.0.weight", "relation_encoder.model.0.bias", "relation_encoder.model.2.weight"):
        big[k] *= 1.0e3                                          # layer-1 outputs ~1e3, layer-2 outputs ~1e6 > 65504
    m = make_model(big, prec="fast")
    assert m.take_status() == 0
    m(*args(), **kw)
    with w.catch_warnings(record=True) as rec:
        w.simplefilter("always")
        assert m.take_status() & 1 and any("fp16" in str(x.message) for x in rec)
    assert m.take_status() == 0                                  # read-and-clear
    m.set_option("precision", 1)
    _, mot = m(*args(), **kw)
    assert m.take_status() == 0 and torch.isfinite(mot).all()
    # a huge per-edge term alone is no overflow any more: W_rp[:, :F] x 1e4 on the seed-0 weights (|Eterm| ~ 1e4 .. 1e5, beyond fp16 with the bias)
    wide = {k: v.copy() for k, v in weights.items()}
    wide["relation_propagator.linear.weight"][:, :150] *= 3.0e4
    wide["relation_propagator.linear.bias"] *= 3.0e4
    m2 = make_model(wide, prec="fast")
    _, mot2 = m2(*args(), **kw)
    m1 = make_model(wide, prec="bf16x3")
    _, mot1 = m1(*args(), **kw)
    assert m2.take_status() == 0 and torch.isfinite(mot2).all()
    assert float((mot2 - mot1).abs().max()) <= 1e-4 * max(1.0, float(mot1.abs().max()))


def test_fast_mode_holds_the_gate_at_large_motions(weights):
    """Until r03 the default mode deviated by ~5e-4 of the largest predicted motion and flagged forwards above 0.125.  The r04 arithmetic has no
    such range: with the decoder scaled x4 and x16 (motions up to ~1.5) and tool actions of +-0.5 the default mode stays within 5e-5 of the
    exact-fp32 engine on the trained rope weights, status 0."""
    tw = load_golden("weights_trained_rope")
    g = synth.make_graph_inputs("rope", 300, 3, seed=5, spacing=0.1)
    n_p = g["n_p"]
    g["action"][:, n_p:] = np.array([[0.5, -0.3, 0.4]], np.float32)
    mm = synth.MATERIALS["rope"]
    csr = aggraph.build_edges(t(g["state"][:, -1]), mm["radius"], t(g["mask"]), t(g["tool_mask"]), mm["topk"], False, "batch", max_tools=1)
    args = (t(g["state"]), t(g["attrs"]), csr, None, t(g["p_instance"]))
    kw = dict(action=t(g["action"]), rope_physics_param=t(g["phys"]))
    for scale in (1.0, 4.0, 16.0):
        exact = make_model(tw, "rope", decoder_scale=scale, prec="f32")
        fast = make_model(tw, "rope", decoder_scale=scale, prec="fast")
        _, ref = exact(*args, **kw)
        _, mot = fast(*args, **kw)
        dev = float((mot - ref).abs().max())
        print(f"decoder x{scale:g}: max |motion| {float(ref.abs().max()):.3f}, fast vs exact-fp32 engine {dev:.2e}")
        assert dev <= 5e-5 * max(1.0, scale / 4.0), dev
        assert fast.take_status() == 0 and exact.take_status() == 0
