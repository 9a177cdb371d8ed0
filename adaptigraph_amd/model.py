"""`DynamicsPredictor` on the HIP engine — drop-in for src/dynamics/gnn/model.py:63-313.

Same constructor, same `forward(state, attrs, Rr, Rs, p_instance, action=..., <material>_physics_param=...)`
returning `(pred_pos, pred_motion)`, same 22-tensor `state_dict()` layout, so
`model.load_state_dict(torch.load("log/<data>/checkpoints/model_100.pth"))` works unchanged
(SURVEY.md §8b).  The torch sub-modules below hold parameters only; all arithmetic runs in
libadaptigraph_hip.so.  Inference only (the north-star path has no backward).
"""
import ctypes
import warnings

import torch
import torch.nn as nn

from . import _lib
from .graph import CSREdges, csr_from_dense, workspace, _stream_ptr, _require_gpu

STATE_DICT_KEYS = (
    [f"particle_encoder.model.{i}.{p}" for i in (0, 2, 4) for p in ("weight", "bias")]
    + [f"relation_encoder.model.{i}.{p}" for i in (0, 2, 4) for p in ("weight", "bias")]
    + [f"particle_propagator.linear.{p}" for p in ("weight", "bias")]
    + [f"relation_propagator.linear.{p}" for p in ("weight", "bias")]
    + [f"non_rigid_predictor.linear_{i}.{p}" for i in (0, 1, 2) for p in ("weight", "bias")]
)


class _MLP3(nn.Module):            # parameter container mirroring Encoder's `model.{0,2,4}` names (model.py:8-15)
    def __init__(self, d_in, d_hidden, d_out):
        super().__init__()
        self.model = nn.Sequential(nn.Linear(d_in, d_hidden), nn.ReLU(), nn.Linear(d_hidden, d_hidden), nn.ReLU(),
                                   nn.Linear(d_hidden, d_out), nn.ReLU())


class _Lin(nn.Module):             # Propagator's `linear` (model.py:27)
    def __init__(self, d_in, d_out):
        super().__init__()
        self.linear = nn.Linear(d_in, d_out)


class _Dec(nn.Module):             # ParticlePredictor's `linear_{0,1,2}` (model.py:47-49)
    def __init__(self, d_in, d_hidden, d_out):
        super().__init__()
        self.linear_0 = nn.Linear(d_in, d_hidden)
        self.linear_1 = nn.Linear(d_hidden, d_hidden)
        self.linear_2 = nn.Linear(d_hidden, d_out)


class DynamicsPredictor(nn.Module):
    def __init__(self, model_config, material_config, dataset_config, device):
        super().__init__()
        self.model_config = model_config
        self.material_config = material_config
        self.dataset_config = dataset_config
        self.device = device
        self.n_his = dataset_config["n_his"]
        self.nf_particle = model_config["nf_particle"]
        self.nf_relation = model_config["nf_relation"]
        self.nf_effect = model_config["nf_effect"]
        self.nf_physics = model_config.get("nf_physics", 0)
        self.eps = 1e-6
        self.motion_clamp = 100

        self.num_materials = len(material_config["material_index"])
        assert self.num_materials == 1, "Only support single material."          # model.py:87-88
        params = material_config[dataset_config["materials"][0]]["physics_params"]
        self.material_dim = sum(1 for p in params if p["use"])                     # model.py:91-94

        for key in ("offset_dim", "density_dim", "state_dim", "rel_density_dim"):
            if model_config.get(key, 0) > 0:
                raise NotImplementedError(f"model_config['{key}'] > 0 is not built into the HIP engine")
        input_dim = model_config["attr_dim"] + model_config["action_dim"] + self.material_dim   # model.py:96-101
        if model_config["rel_particle_dim"] == -1:                                 # model.py:106-107 (mutates, as ref)
            model_config["rel_particle_dim"] = input_dim
        if model_config["rel_particle_dim"] != 0:
            raise NotImplementedError("rel_particle_dim != 0 is not built into the HIP engine")
        assert model_config["rel_group_dim"] == 1 and model_config["rel_distance_dim"] == 3
        rel_input_dim = (model_config["rel_attr_dim"] * 2 + model_config["rel_group_dim"]
                         + model_config["rel_distance_dim"] * self.n_his)          # model.py:109-113
        assert self.nf_particle == self.nf_relation == self.nf_effect

        nf = self.nf_effect
        self.particle_encoder = _MLP3(input_dim, self.nf_particle, nf)
        self.relation_encoder = _MLP3(rel_input_dim, self.nf_relation, nf)
        self.particle_propagator = _Lin(nf * 2, nf)
        self.relation_propagator = _Lin(nf * 3, nf)
        self.non_rigid_predictor = _Dec(nf, nf, 3)

        self._handle = None
        self._sig = None
        _lib.lib()   # fail at construction time, loudly, if the HIP library is absent

    # ------------------------------------------------------------------ weights -> packed device streams
    def _signature(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def _sync_weights(self):
        sig = self._signature()
        if self._handle is not None and sig == self._sig:
            return
        L = _lib.lib()
        sd = self.state_dict()
        host = [sd[k].detach().to("cpu", torch.float32).contiguous() for k in STATE_DICT_KEYS]
        arr = (ctypes.c_void_p * len(host))(*[t.data_ptr() for t in host])
        if self._handle is None:
            cfg = _lib.ModelConfig(self.nf_effect, self.n_his, self.model_config["attr_dim"], self.material_dim,
                                   self.model_config["action_dim"], self.model_config["pstep"],
                                   float(self.motion_clamp))
            h = ctypes.c_void_p()
            _lib.check(L.ag_model_create(ctypes.byref(cfg), arr, ctypes.byref(h)), "ag_model_create")
            self._handle = h
        else:
            _lib.check(L.ag_model_update_weights(self._handle, arr), "ag_model_update_weights")
        self._sig = sig

    def handle(self, device):
        with torch.cuda.device(device):
            self._sync_weights()
        return self._handle

    def take_status(self, device=None):
        """Read-and-clear the model's sticky numeric status (ag_model_status; synchronises the current stream).
        Bit 0: a forward left the range of its arithmetic — in precision mode 2 an fp16 activation of the edge stack beyond
        65504 or a non-finite per-edge term, in any mode non-finite inputs (include/adaptigraph_hip.h).  Warns once per
        occurrence and returns the flag word."""
        dev = torch.device(device if device is not None else self.device)
        flags = ctypes.c_int(0)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().ag_model_status(self.handle(dev), ctypes.byref(flags), _stream_ptr(dev)), "ag_model_status")
        if flags.value & 1:
            warnings.warn("adaptigraph_amd: a forward produced non-finite values. If the inputs were finite, an activation of the fp16 "
                          "edge stack of precision mode 2 ('fast') left fp16's range (|x| > 65504): use "
                          "model.set_option('precision', 1).", RuntimeWarning, stacklevel=2)
        return flags.value

    def set_option(self, name, value, device=None):
        """Engine knob (include/adaptigraph_hip.h: ag_set_option): "precision" 0 = exact fp32 MFMA, 1 = split-bf16 with an
        fp32 per-edge table, 2 = split-bf16 node stacks + fp16 edge stack with residual bytes + 16-bit block-scaled per-edge
        table ("fast", the default; its fp16 activations have fp16's range, see take_status); "rollout_streams"; "node_dedup";
        "fuse_aggregate"; "max_blocks"; "edge_products"; "edge_stationary"; "node_stationary"; "cu_split"; "self_edges" (r06: self-loops of the
        attribute classes (1, 0) / (0, 1) as one table row per class, default 1); "shared_state" (r06: rollouts of one cloud under many sampled
        pushes compute per sample only what can differ from the tool-less base trajectory, default 0).  All of them give the same bits, except
        "precision" and "agg_q16" (r06, default 0: the per-node sums of a round travel as 16-bit block-scaled rows between the segment reduce and
        node_update in mode 2 — one more rounding per node and round)."""
        dev = torch.device(device if device is not None else self.device)
        _lib.check(_lib.lib().ag_set_option(self.handle(dev), name.encode(), int(value)), f"ag_set_option({name})")
        return self

    def get_option(self, name, device=None):
        """The engine's CURRENT value of an option (ag_get_option: default, environment or set_option)."""
        import ctypes
        dev = torch.device(device if device is not None else self.device)
        v = ctypes.c_int(0)
        _lib.check(_lib.lib().ag_get_option(self.handle(dev), name.encode(), ctypes.byref(v)), f"ag_get_option({name})")
        return v.value

    def __del__(self):
        try:
            if self._handle is not None:
                _lib.lib().ag_model_destroy(self._handle)
        except Exception:
            pass

    # ------------------------------------------------------------------ forward (model.py:129-313)
    @torch.no_grad()
    def forward(self, state, attrs, Rr, Rs, p_instance, action=None, particle_den=None, obj_mask=None, **kwargs):
        _require_gpu(state, "state")
        dev = state.device
        B, N = attrs.size(0), attrs.size(1)
        n_p, n_inst = p_instance.size(1), p_instance.size(2)
        physics_keys = [k for k in kwargs.keys() if k.endswith("_physics_param")]
        assert len(physics_keys) == 1                                              # model.py:184-185
        phys = kwargs[physics_keys[0]].to(dev, torch.float32).contiguous()
        assert phys.shape == (B, self.material_dim)
        assert action is not None                                                  # model.py:193-194
        assert state.shape == (B, self.n_his, N, 3)
        edges = Rr if isinstance(Rr, CSREdges) else csr_from_dense(Rr, Rs)
        assert edges.B == B and edges.N == N

        state = state.contiguous().float()
        attrs = attrs.contiguous().float()
        action = action.to(dev).contiguous().float()
        p_instance = p_instance.contiguous().float()
        pred_pos = torch.empty((B, n_p, 3), dtype=torch.float32, device=dev)
        pred_motion = torch.empty((B, n_p, 3), dtype=torch.float32, device=dev)
        L = _lib.lib()
        h = self.handle(dev)
        ws = workspace(dev, L.ag_forward_workspace_bytes_for(h, B, N, edges.e_cap))
        with torch.cuda.device(dev):
            rc = L.ag_forward(h, state.data_ptr(), attrs.data_ptr(), action.data_ptr(), p_instance.data_ptr(),
                              n_inst, phys.data_ptr(), edges.row_ptr.data_ptr(), edges.edge_recv.data_ptr(),
                              edges.edge_send.data_ptr(), edges.e_cap, B, N, n_p, pred_pos.data_ptr(),
                              pred_motion.data_ptr(), ws.data_ptr(), ws.numel(), _stream_ptr(dev))
        _lib.check(rc, "ag_forward")
        return pred_pos, pred_motion
