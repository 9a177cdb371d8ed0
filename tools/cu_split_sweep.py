"""CU-partitioned rollout (ag_set_option("cu_split", X)): throughput of dynamics() at a bench shape for a sweep of X (CUs of the MFMA partition),
each checked bit for bit against the unpartitioned result.

    python tools/cu_split_sweep.py [material] [batch] [rollout steps] [parts] [X,X,...] > profiles/r05_cu_split_sweep_<material>.txt
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                            # noqa: E402
from adaptigraph_amd import synth                                       # noqa: E402
from adaptigraph_amd.forward_dynamics import dynamics                   # noqa: E402

material = sys.argv[1] if len(sys.argv) > 1 else "rope"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
T = int(sys.argv[3]) if len(sys.argv) > 3 else 10
parts = int(sys.argv[4]) if len(sys.argv) > 4 else 2
splits = [int(x) for x in sys.argv[5].split(",")] if len(sys.argv) > 5 else [0, 64, 80, 88, 96, 104, 112, 120, 128, 144, 160]
dev = "cuda:0"
weights = dict(np.load(os.path.join(ROOT, "tests", "golden", "weights_seed0.npz")))
eng = bench.Engine(material, weights, dev, 1)
wl = bench.WORKLOADS[material]
state_np, act_np = synth.make_mpc_inputs(material, wl["n_obj"], batch, seed=0, len_lo=T, len_hi=T + 0.9, **wl["kw"])
state, action = torch.from_numpy(state_np).to(dev), torch.from_numpy(act_np).to(dev)
eng.opt("rollout_streams", parts)
print(f"# {material} {wl['n_obj']} particles, batch {batch}, {T}-step rollout, {parts} batch parts; cu_split = CUs of the MFMA partition (0 = off: streams share the chip)")
print("# cu_split | ms per rollout | graph-steps/s | bitwise vs cu_split 0 | model_status")
ref = None
for x in splits:
    eng.opt("cu_split", x)
    for _ in range(3):
        out = dynamics(state, action, eng.model, dev, eng.ppm)["state_seqs"]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        out = dynamics(state, action, eng.model, dev, eng.ppm)["state_seqs"]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    st = eng.model.take_status()
    if ref is None:
        ref = out.clone()
    # per-kernel-class launch times WHILE the partitions co-run (HIP events on the launching queue)
    import ctypes
    from adaptigraph_amd import _lib
    L, h = _lib.lib(), eng.h
    L.ag_profile_enable(h, 1)
    dynamics(state, action, eng.model, dev, eng.ppm)
    torch.cuda.synchronize()
    ms = (ctypes.c_double * 6)(); cnt = (ctypes.c_int64 * 6)(); e = ctypes.c_int64()
    L.ag_profile_read(h, ms, cnt, ctypes.byref(e))
    L.ag_profile_enable(h, 0)
    k = " ".join(f"{n}={ms[i] / max(cnt[i], 1):.3f}" for i, n in enumerate(_lib.KERNEL_CLASSES) if cnt[i])
    print(f"{x:5d} | {dt * 1e3:8.3f} | {batch * T / dt:10.0f} | {bool(torch.equal(out, ref))} | {st} | {k}", flush=True)
