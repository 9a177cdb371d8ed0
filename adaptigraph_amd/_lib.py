"""ctypes binding of libadaptigraph_hip.so (C ABI in include/adaptigraph_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or fails to load, importing
anything that computes raises immediately.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AG_LIB_PATH") or os.path.join(_HERE, "libadaptigraph_hip.so")   # override: kernel A/B builds
CSRC = os.path.join(_HERE, "csrc")

EXPORTS = ("ag_last_error", "ag_version", "ag_model_create", "ag_model_update_weights", "ag_model_destroy",
           "ag_edge_capacity", "ag_edges_workspace_bytes", "ag_build_edges", "ag_forward_workspace_bytes",
           "ag_forward", "ag_rollout_workspace_bytes", "ag_rollout", "ag_profile_enable", "ag_profile_read", "ag_set_option", "ag_chamfer", "ag_chamfer_masked", "ag_gather_rows", "ag_segment_sum",
           "ag_message_forward", "ag_message_backward", "ag_model_status", "ag_train_pack", "ag_train_chain", "ag_train_weight_grads", "ag_train_weight_grads_workspace_bytes", "ag_add3_relu", "ag_relu_mask", "ag_train_weight_grads_into", "ag_edge_inputs_forward", "ag_edge_inputs_backward",
           "ag_forward_workspace_bytes_for", "ag_rollout_workspace_bytes_for", "ag_rollout_streams_for", "ag_get_option")
KERNEL_CLASSES = ("build_edges", "node_encode", "edge_encode", "aggregate", "node_update", "rollout_step")

AG_VARIANT_SINGLE, AG_VARIANT_BATCH = 0, 1
AG_HEIGHT_MIN, AG_HEIGHT_MASKED_MEAN = 0, 1

c_void_p, c_int, c_int64, c_size_t, c_float = (ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_size_t,
                                               ctypes.c_float)


class ModelConfig(ctypes.Structure):
    _fields_ = [("nf", ctypes.c_int32), ("n_his", ctypes.c_int32), ("attr_dim", ctypes.c_int32),
                ("phys_dim", ctypes.c_int32), ("action_dim", ctypes.c_int32), ("pstep", ctypes.c_int32),
                ("motion_clamp", ctypes.c_float)]


class RolloutParams(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int32), ("N", ctypes.c_int32), ("n_p", ctypes.c_int32), ("n_instance", ctypes.c_int32),
                ("topk", ctypes.c_int32), ("connect_tools_all", ctypes.c_int32), ("max_tools", ctypes.c_int32),
                ("n_steps", ctypes.c_int32), ("height_mode", ctypes.c_int32), ("gripper_raise", ctypes.c_float)]


def build(force=False):
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")) or f in ("Makefile", "exports.map")]
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "adaptigraph_hip.h"))
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", CSRC, "-j8"] + (["-B"] if force else []), stdout=subprocess.DEVNULL)
    return LIB_PATH


_LIB = None


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"adaptigraph_amd: {LIB_PATH} is missing. The engine has no CPU fallback; build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc, no GPU required).")
    # Load torch (and with it torch's bundled libamdhip64.so.7) FIRST: device pointers, streams and events are
    # shared with torch, so both must sit on ONE HIP runtime instance; the loader then binds our DT_NEEDED
    # libamdhip64.so.7 to the already-loaded copy.  (Loaded the other way round, torch finds no device.)
    import torch  # noqa: F401
    L = ctypes.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(L, name):
            raise RuntimeError(f"adaptigraph_amd: {LIB_PATH} does not export {name} (stale build?)")
    L.ag_last_error.restype = ctypes.c_char_p
    L.ag_version.restype = c_int
    L.ag_model_create.restype = c_int
    L.ag_model_create.argtypes = [ctypes.POINTER(ModelConfig), ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p)]
    L.ag_model_update_weights.restype = c_int
    L.ag_model_update_weights.argtypes = [c_void_p, ctypes.POINTER(c_void_p)]
    L.ag_model_destroy.restype = c_int
    L.ag_model_destroy.argtypes = [c_void_p]
    L.ag_edge_capacity.restype = c_int64
    L.ag_edge_capacity.argtypes = [c_int] * 5
    L.ag_edges_workspace_bytes.restype = c_size_t
    L.ag_edges_workspace_bytes.argtypes = [c_int] * 5
    L.ag_build_edges.restype = c_int
    L.ag_build_edges.argtypes = [c_void_p] * 4 + [c_int] * 6 + [c_void_p] * 3 + [c_int64, c_void_p, c_size_t, c_void_p]
    L.ag_forward_workspace_bytes.restype = c_size_t
    L.ag_forward_workspace_bytes.argtypes = [c_int, c_int, c_int64]
    L.ag_forward_workspace_bytes_for.restype = c_size_t
    L.ag_forward_workspace_bytes_for.argtypes = [c_void_p, c_int, c_int, c_int64]
    L.ag_rollout_workspace_bytes_for.restype = c_size_t
    L.ag_rollout_workspace_bytes_for.argtypes = [c_void_p, ctypes.POINTER(RolloutParams)]
    L.ag_rollout_streams_for.restype = c_int
    L.ag_rollout_streams_for.argtypes = [c_void_p, ctypes.POINTER(RolloutParams)]
    L.ag_forward.restype = c_int
    L.ag_forward.argtypes = ([c_void_p] * 5 + [c_int] + [c_void_p] * 4 + [c_int64] + [c_int] * 3 + [c_void_p] * 3 +
                             [c_size_t, c_void_p])
    L.ag_rollout_workspace_bytes.restype = c_size_t
    L.ag_rollout_workspace_bytes.argtypes = [ctypes.POINTER(RolloutParams)]
    L.ag_rollout.restype = c_int
    L.ag_rollout.argtypes = [c_void_p, ctypes.POINTER(RolloutParams)] + [c_void_p] * 13 + [c_size_t, c_void_p]
    L.ag_chamfer.restype = c_int
    L.ag_chamfer.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]
    L.ag_gather_rows.restype = c_int
    L.ag_gather_rows.argtypes = [c_void_p, c_void_p, c_void_p, ctypes.c_int64, c_int, c_void_p]
    L.ag_segment_sum.restype = c_int
    L.ag_segment_sum.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_int64, c_int, c_void_p]
    L.ag_message_forward.restype = c_int
    L.ag_message_forward.argtypes = [c_void_p] * 6 + [ctypes.c_int64, c_int, c_void_p]
    L.ag_message_backward.restype = c_int
    L.ag_message_backward.argtypes = [c_void_p] * 8 + [ctypes.c_int64, c_int, c_void_p]
    L.ag_chamfer_masked.restype = c_int
    L.ag_chamfer_masked.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]
    L.ag_train_pack.restype = c_int
    L.ag_train_pack.argtypes = [c_void_p, c_void_p] + [c_int] * 8 + [c_void_p, c_void_p]
    L.ag_train_chain.restype = c_int
    L.ag_train_chain.argtypes = [c_int, c_int, c_int, c_void_p, c_void_p, ctypes.POINTER(c_void_p), c_void_p, ctypes.POINTER(c_void_p), c_void_p,
                                 ctypes.c_int64, c_int, c_void_p]
    L.ag_train_weight_grads_workspace_bytes.restype = c_size_t
    L.ag_train_weight_grads_workspace_bytes.argtypes = [ctypes.c_int64, c_int]
    L.ag_train_weight_grads.restype = c_int
    L.ag_train_weight_grads.argtypes = [c_int, ctypes.POINTER(c_void_p), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(c_void_p),
                                        ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32), ctypes.c_int64, c_void_p, c_void_p, c_size_t,
                                        c_void_p]
    L.ag_train_weight_grads_into.restype = c_int
    L.ag_train_weight_grads_into.argtypes = [c_int, ctypes.POINTER(c_void_p), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(c_void_p),
                                             ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32), ctypes.c_int64, c_void_p,
                                             ctypes.POINTER(c_void_p), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(c_void_p),
                                             ctypes.POINTER(ctypes.c_int32), c_void_p, c_size_t, c_void_p]
    L.ag_edge_inputs_forward.restype = c_int
    L.ag_edge_inputs_forward.argtypes = [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, ctypes.c_int64, c_void_p]
    L.ag_edge_inputs_backward.restype = c_int
    L.ag_edge_inputs_backward.argtypes = [c_void_p, c_int, c_int, c_int] + [c_void_p] * 9 + [ctypes.c_int64, ctypes.c_int64, c_void_p]
    L.ag_add3_relu.restype = c_int
    L.ag_add3_relu.argtypes = [c_void_p] * 4 + [ctypes.c_int64, c_void_p]
    L.ag_relu_mask.restype = c_int
    L.ag_relu_mask.argtypes = [c_void_p] * 3 + [ctypes.c_int64, c_void_p]
    L.ag_model_status.restype = c_int
    L.ag_model_status.argtypes = [c_void_p, ctypes.POINTER(c_int), c_void_p]
    L.ag_set_option.restype = c_int
    L.ag_set_option.argtypes = [c_void_p, ctypes.c_char_p, c_int]
    L.ag_get_option.restype = c_int
    L.ag_get_option.argtypes = [c_void_p, ctypes.c_char_p, ctypes.POINTER(c_int)]
    L.ag_profile_enable.restype = c_int
    L.ag_profile_enable.argtypes = [c_void_p, c_int]
    L.ag_profile_read.restype = c_int
    L.ag_profile_read.argtypes = [c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_int64), ctypes.POINTER(c_int64)]
    _LIB = L
    return L


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {lib().ag_last_error().decode()}")
