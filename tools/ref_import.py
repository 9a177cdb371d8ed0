"""Import the AdaptiGraph reference (read-only, /root/reference) in THIS container.

Only used by tools/gen_golden.py to produce tests/golden/*.npz.  It never ships:
the GPU box has no /root/reference, and nothing under tests/ or the package
imports this module.

The hot-path modules import third-party packages at module import time that the
hot functions never call (dgl, cv2, moviepy, h5py); those are replaced by empty
module stubs so that the reference's own code runs unmodified.
"""
import sys
import types

REF_SRC = "/root/reference/src"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference():
    sys.dont_write_bytecode = True
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)
    if "dgl" not in sys.modules:
        dgl = _stub("dgl")
        dgl.geometry = _stub("dgl.geometry", farthest_point_sampler=None)
    for n in ("cv2", "moviepy", "moviepy.editor", "h5py"):
        if n not in sys.modules:
            _stub(n)
    from dynamics.gnn.model import DynamicsPredictor
    from dynamics.dataset.graph import (construct_edges_from_states,
                                        construct_edges_from_states_batch)
    from dynamics.utils import truncate_graph, pad_torch
    from planning.forward_dynamics import dynamics, dynamics_masked
    from planning.plan_utils import decode_action
    # planning glue ("next" row n1): plan.py pulls in robot / perception / open3d modules at import time that
    # running_cost never touches -> stub them; planner.py and losses.py import cleanly.
    for n in ("open3d", "planning.real_world.real_env", "planning.perception"):
        if n not in sys.modules:
            _stub(n, RealEnv=None, PerceptionModule=None, get_state_cur=None)
    # physics_param_optimizer.py imports cma / skopt (absent) for its black-box search; dynamics_error (row n2) needs neither
    for n in ("cma", "skopt", "skopt.learning", "skopt.learning.gaussian_process", "skopt.learning.gaussian_process.kernels",
              "skopt.utils"):
        if n not in sys.modules:
            _stub(n, gp_minimize=None, WhiteKernel=None, RBF=None, Matern=None, GaussianProcessRegressor=None,
                  expected_minimum=None)
    from planning import losses, plan_utils, physics_param_optimizer
    from planning.plan import running_cost
    from planning.real_world.planner import Planner
    return types.SimpleNamespace(
        losses=losses, plan_utils=plan_utils, running_cost=running_cost, Planner=Planner,
        physics_param_optimizer=physics_param_optimizer,
        DynamicsPredictor=DynamicsPredictor,
        construct_edges_from_states=construct_edges_from_states,
        construct_edges_from_states_batch=construct_edges_from_states_batch,
        truncate_graph=truncate_graph, pad_torch=pad_torch,
        dynamics=dynamics, dynamics_masked=dynamics_masked,
        decode_action=decode_action)
