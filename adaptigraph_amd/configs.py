"""The reference's shipped configuration values for the hot path, as plain dicts.

These are the keys the engine reads from the dicts a reference user already has
(`yaml.safe_load` of src/config/dynamics/<m>.yaml and src/config/planning/<m>.yaml): they are
accepted unchanged by `DynamicsPredictor(model_config, material_config, dataset_config, device)`
and by `dynamics(..., ppm_optimizer)`.  Only hot-path keys are listed (SURVEY.md §5 "config").
"""
import copy
import types

MODEL_CONFIG = dict(   # config/dynamics/rope.yaml:55-78 (identical for granular, cloth)
    verbose=False, nf_particle=150, nf_relation=150, nf_effect=150, nf_physics=10,
    attr_dim=2, state_dim=0, offset_dim=0, action_dim=3, density_dim=0, pstep=3, sequence_len=4,
    rel_particle_dim=0, rel_attr_dim=2, rel_group_dim=1, rel_distance_dim=3, rel_density_dim=0)

_USED_PARAM = dict(rope="stiffness", granular="granular_scale", cloth="sf")   # the one `use: True` entry


def model_config():
    return copy.deepcopy(MODEL_CONFIG)


def material_config(material):
    """material_config with exactly one used physics parameter (rope.yaml:109-112, granular.yaml:101-104)."""
    return {"material_index": {material: 0},
            material: {"physics_params": [{"name": "particle_radius", "use": False, "min": 0.0, "max": 1.0},
                                          {"name": _USED_PARAM[material], "use": True, "min": 0.0, "max": 1.0}]}}


def dataset_config(material, n_his=4):
    return {"data_name": material, "materials": [material], "n_his": n_his, "n_future": 3}


_TASK_COMMON = dict(sim_real_ratio=10, max_n=1, max_nobj=200, max_nR=2000, n_his=4, n_look_ahead=1, noise_level=1.0,
                    reward_weight=500.0)
TASK_CONFIG = {   # config/planning/{rope,granular,cloth}.yaml
    "rope": dict(_TASK_COMMON, material="rope", material_indices={"rope": 0}, material_dims={"rope": 1},
                 adj_thresh=0.50, fps_radius=0.20, eef_num=1, topk=10, connect_tools_all=False, push_length=0.1,
                 pusher_points=[[0.0, 0.0, 0.12]], gripper_enable=False,
                 action_lower_lim=[-4.5, -2.5, -3.14, 5], action_upper_lim=[0.0, 4.5, 3.14, 15]),
    "granular": dict(_TASK_COMMON, material="granular", material_indices={"granular": 0},
                     material_dims={"granular": 1}, adj_thresh=0.40, fps_radius=0.20, eef_num=5,
                     topk=20, connect_tools_all=False,
                     push_length=0.2, gripper_enable=False,
                     pusher_points=[[0.0, 0.0, 0.1], [0.0, 0.05, 0.1], [0.0, 0.025, 0.1], [0.0, -0.025, 0.1],
                                    [0.0, -0.05, 0.1]],
                     action_lower_lim=[-4.5, -2.5, -3.14, 2], action_upper_lim=[0.0, 4.5, 3.14, 10]),
    "cloth": dict(_TASK_COMMON, material="cloth", material_indices={"cloth": 0}, material_dims={"cloth": 1},
                  adj_thresh=0.75, fps_radius=0.30, eef_num=1, topk=5, connect_tools_all=True, push_length=0.1,
                  pusher_points=[[0.0, 0.0, 0.170]], gripper_enable=True,
                  action_lower_lim=[-4.5, -2.5, -3.14, 2], action_upper_lim=[0.0, 4.5, 3.14, 10]),
}


def task_config(material):
    return copy.deepcopy(TASK_CONFIG[material])


def ppm_optimizer_stub(material, physics_param=None, adj_thresh=None):
    """Stand-in for planning.physics_param_optimizer.PhysicsParamOnlineOptimizer carrying exactly the
    attributes dynamics()/dynamics_masked() read (forward_dynamics.py:14-18,27,30,117-120)."""
    task = task_config(material)
    ns = types.SimpleNamespace()
    ns.task_config = task
    ns.eef_num = task["eef_num"]
    ns.material = material
    ns.material_dims = task["material_dims"]
    ns.material_indices = task["material_indices"]
    ns.adj_thresh = task["adj_thresh"] if adj_thresh is None else adj_thresh
    ns.physics_param = physics_param   # {material: tensor(material_dim)}; filled by the caller (needs torch)
    return ns
