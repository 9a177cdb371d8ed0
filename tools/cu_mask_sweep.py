"""How do the step's kernels scale with the CUs they get?  (can the MFMA-bound edge encoder and the HBM-bound
segment reduce / node update overlap by CU partitioning?)

One C2 forward (rope-1k x 256, default mode) on a stream created with hipExtStreamCreateWithCUMask, for CU counts 32 ... 256 and two
mask patterns ("first": bits [0, n), "last": bits [256 - n, 256) — mask bit b is CU slot b / 8 of XCD b % 8, tools/ubench/cu_mask_map.hip, so both
are n / 8 CUs of every XCD; a mask that leaves an XCD's slice empty lets that XCD run unrestricted), per-kernel-class launch time from the library's
own HIP events (ag_profile_*).  If the reduce's time is ~ 1 / CUs, then giving it the CUs the edge encoder does not take cannot beat
running the two one after the other on all CUs.

    python tools/cu_mask_sweep.py [reps] > profiles/r05_cu_mask_sweep.txt
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptigraph_amd import _lib, configs, synth                       # noqa: E402
from adaptigraph_amd.graph import build_edges                          # noqa: E402
from adaptigraph_amd.model import DynamicsPredictor                    # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = "cuda:0"


def hip_runtime():
    """The HIP runtime instance torch already loaded (streams must come from the same one)."""
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line:
            return ctypes.CDLL(line.split()[-1])
    raise RuntimeError("libamdhip64 not mapped")


def masked_stream(hip, bits):
    words = (ctypes.c_uint32 * 8)()
    for b in bits:
        words[b >> 5] |= 1 << (b & 31)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, f"hipExtStreamCreateWithCUMask -> {rc}"
    return s


w = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "weights_seed0.npz")))
m = DynamicsPredictor(configs.model_config(), configs.material_config("rope"), configs.dataset_config("rope"), dev)
m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
m = m.to(dev).eval()
g = synth.make_graph_inputs("rope", 1000, 256, seed=0, spacing=0.1)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)      # noqa: E731
csr = build_edges(t(g["state"][:, -1]), 0.5, t(g["mask"]), t(g["tool_mask"]), 10, False, "batch", max_tools=1)
args = (t(g["state"]), t(g["attrs"]), csr, None, t(g["p_instance"]))
kw = dict(action=t(g["action"]), rope_physics_param=t(g["phys"]))
ref = m(*args, **kw)[1].clone()
torch.cuda.synchronize()
L = _lib.lib()
h = m.handle(torch.device(dev))
hip = hip_runtime()
n_cu = torch.cuda.get_device_properties(0).multi_processor_count
E = int(csr.row_ptr[-1].item())
n_nodes = 256 * 1001
agg_bytes = E * 320 + n_nodes * (1.0 + 2.5 + 2.5) / 3.0 * 640          # bench.py's roofline_hbm accounting (average of the three rounds)
print(f"# C2 forward (rope 1000+1 x 256, E = {E}), default mode, {reps} forwards per row; device reports {n_cu} CUs")
print("# pattern  CUs | edge_encode ms | aggregate ms (avg of 3 rounds)  TB/s  x CUs/256 | node_update ms (avg of 3 rounds) | bitwise")
for pattern in ("first", "last"):
    for n in (32, 64, 96, 128, 160, 192, 224, 256):
        bits = list(range(n)) if pattern == "first" else list(range(256 - n, 256))
        s = masked_stream(hip, bits)
        ext = torch.cuda.ExternalStream(s.value)
        with torch.cuda.stream(ext):
            for _ in range(2):
                out = m(*args, **kw)[1]
            ext.synchronize()
            L.ag_profile_enable(h, 1)
            for _ in range(reps):
                out = m(*args, **kw)[1]
            ext.synchronize()
            ms = (ctypes.c_double * 6)()
            cnt = (ctypes.c_int64 * 6)()
            e = ctypes.c_int64()
            L.ag_profile_read(h, ms, cnt, ctypes.byref(e))
            L.ag_profile_enable(h, 0)
        k = {nm: ms[i] / max(cnt[i], 1) for i, nm in enumerate(_lib.KERNEL_CLASSES)}
        tbs = agg_bytes / (k["aggregate"] * 1e-3) / 1e12
        print(f"{pattern:8s} {n:4d} | {k['edge_encode']:7.4f} | {k['aggregate']:7.4f}  {tbs:5.2f}  {tbs * 256 / n:5.2f} | {k['node_update']:7.4f} | "
              f"{bool(torch.equal(out, ref))}")
        hip.hipStreamDestroy(s)
