#!/usr/bin/env python3
"""Headline benchmark: rollout graph-steps/s (batch x steps / wall) on N MI355X — BASELINE.json `metric`.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: `dynamics()` on `--batch` action samples per GPU
(default 256) of the rope-1k cloud with a `--rollout-steps`-step (default 10) push, i.e. BASELINE configs[1]
"rope, ~1k particles, batch 256, 10-step rollout": per model step a radius-graph rebuild, the GNN forward and
the state/tool update for every graph.  Multi-GPU: the batch shards across ranks with no data-path collective
(weak scaling: per-GPU batch fixed), then ONE RCCL all-gather of the predicted states (north_star).
Prints one JSON line on rank 0 (contract in the task statement), including `roofline` for the dominant
kernel (edge_encode_kernel, fp32 MFMA bound) timed live with HIP events on the launch stream, and
`cpu_baseline` = the CPU oracle ("port") timed on a bounded sample of the same workload on this box's cores.
"""
import argparse
import ctypes
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on these hosts (RCCL across processes needs it)

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from adaptigraph_amd import _lib, configs, synth                       # noqa: E402
from adaptigraph_amd import dist as agdist                             # noqa: E402
from adaptigraph_amd.forward_dynamics import dynamics                  # noqa: E402
from adaptigraph_amd.model import DynamicsPredictor                    # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0       # MI355X_MICROARCH.md "Peak BF16/FP16 MFMA" (dense)
PEAK_HBM_GBS = 8000.0                # HBM3E spec; 6290 GB/s is the measured float4-copy ceiling (same file)
FLOP_PER_EDGE = 2 * (17 * 150 + 3 * 150 * 150)   # edge encoder 17->150->150->150 + W_rp[:, :150] block
PRECISIONS = {"f32": 0, "bf16x3": 1, "fast": 2}
# measured HBM traffic per launch, bytes (KB counters x 1024): edge_encode 2 x 27 176 KB read + 782 653 KB written;
# aggregate_half 2 x 555 526 KB read + 160 160 KB written
PMC_TRAFFIC = {("rope", 256, "fast", "edge_encode"): (2 * 27175.8 + 782652.8) * 1024,
               ("rope", 256, "fast", "aggregate"): (2 * 555526.0 + 160160.4) * 1024}
DTYPE = {"f32": "f32 (exact fp32 MFMA)",
         "bf16x3": "f32 operands split hi+lo bf16, 3 bf16 MFMAs per product, f32 accumulate",
         "fast": "f32 operands split hi+lo bf16, 3 bf16 MFMAs per product, f32 accumulate; per-edge table stored f16"}
WORKLOADS = {"rope": dict(n_obj=1000, kw=dict(spacing=0.1)), "granular": dict(n_obj=2000, kw={}),
             "cloth": dict(n_obj=4096, kw={})}


def cpu_baseline(weights, material, n_obj, kw, rollout_steps, seconds_budget=20.0):
    """Oracle ("port" of the reference algorithm: dense-formulation forward, O(N^2) edge build, per-step rebuild)
    on a bounded sample: as many graphs as host threads, 2 rollout steps."""
    from oracle import ag_oracle as ago
    cores = os.cpu_count() or 1
    bsz = max(1, min(cores, 32))
    steps = 2
    state, act = synth.make_mpc_inputs(material, n_obj, bsz, seed=0, len_lo=steps, len_hi=steps + 0.9, **kw)
    task = configs.task_config(material)
    ago.lib()
    t0 = time.perf_counter()
    reps = 0
    while True:
        ago.dynamics(weights, task, state, act)
        reps += 1
        dt = time.perf_counter() - t0
        if dt > seconds_budget * 0.5:
            break
    return {"value": bsz * steps * reps / dt, "unit": "graph-steps/s", "cores": cores, "kind": "port",
            "sample": f"{material} n_obj={n_obj}, batch {bsz}, {steps}-step rollout x {reps} reps ({dt:.1f} s), "
                      f"OpenMP over graphs, {cores} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--material", default="rope", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=256, help="action samples (graphs) per GPU")
    ap.add_argument("--rollout-steps", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel HIP-event roofline pass")
    ap.add_argument("--precision", default="fast", choices=sorted(PRECISIONS),
                    help="engine arithmetic mode; all three pass the 1e-4 parity gate (tests/test_gpu_parity.py)")
    ap.add_argument("--streams", type=int, default=2, help="rollout batch parts on separate streams (engine default 2)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (the engine has no CPU fallback)"
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device(dev))     # "nccl" is RCCL on ROCm

    wl = WORKLOADS[args.material]
    weights = dict(np.load(os.path.join(ROOT, "tests", "golden", "weights_seed0.npz")))
    model = DynamicsPredictor(configs.model_config(), configs.material_config(args.material),
                              configs.dataset_config(args.material), dev)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})
    model = model.to(dev).eval()
    ppm = configs.ppm_optimizer_stub(args.material)
    ppm.physics_param = {args.material: torch.tensor([0.5], device=dev)}

    T = args.rollout_steps
    B_global = args.batch * world
    state_np, act_np = synth.make_mpc_inputs(args.material, wl["n_obj"], B_global, seed=0, len_lo=T, len_hi=T + 0.9,
                                             **wl["kw"])
    state = torch.from_numpy(state_np).to(dev)           # inputs resident in HBM before the timed region
    action = torch.from_numpy(act_np).to(dev)

    def one_pass():
        return agdist.dynamics_sharded(dynamics, state, action, model, dev, ppm)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    L = _lib.lib()
    h = model.handle(torch.device(dev))

    def set_opt(name, value):
        _lib.check(L.ag_set_option(h, name.encode(), int(value)), f"ag_set_option({name})")

    set_opt("precision", PRECISIONS[args.precision])
    set_opt("rollout_streams", args.streams)
    for _ in range(args.warmup):
        out = one_pass()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_pass()
    sync()
    dt = time.perf_counter() - t0
    assert out["state_seqs"].shape == (B_global, 1, wl["n_obj"], 3) and bool(torch.isfinite(out["state_seqs"]).all())

    # Roofline pass (outside the timed region): the same workload with the rollout on ONE stream, so every kernel
    # has the GPU to itself and a launch duration means what a roofline needs it to mean (in the timed region two
    # half-batch streams co-run and a kernel's wall time includes its neighbour's share of the CUs).
    roof = None
    kernels = None
    if not args.no_profile:
        set_opt("rollout_streams", 1)
        one_pass()
        _lib.check(L.ag_profile_enable(h, 1), "ag_profile_enable")
        n_prof = max(1, min(3, args.steps))
        for _ in range(n_prof):
            one_pass()
        ms = (ctypes.c_double * 6)()
        cnt = (ctypes.c_int64 * 6)()
        edges = ctypes.c_int64()
        _lib.check(L.ag_profile_read(h, ms, cnt, ctypes.byref(edges)), "ag_profile_read")
        _lib.check(L.ag_profile_enable(h, 0), "ag_profile_enable")
        set_opt("rollout_streams", args.streams)
        kernels = {name: {"ms_per_launch": ms[i] / max(int(cnt[i]), 1), "launches": int(cnt[i])}
                   for i, name in enumerate(_lib.KERNEL_CLASSES)}
        k = _lib.KERNEL_CLASSES.index("edge_encode")
        roof_hbm = None
        if cnt[k] > 0 and ms[k] > 0:
            avg_s = ms[k] / cnt[k] * 1e-3
            e_per = edges.value / cnt[k]
            b3 = args.precision != "f32"
            # algorithmic FLOP of the kernel's arithmetic: the split-bf16 modes need 3 bf16 products per fp32 product
            flop_edge = FLOP_PER_EDGE * (3 if b3 else 1)
            peak = PEAK_BF16_MFMA_TFLOPS if b3 else PEAK_FP32_MFMA_TFLOPS
            achieved = flop_edge * e_per / avg_s / 1e12
            # HBM bytes per launch from the PMC passes committed in profiles/r01_final_traffic_1stream.txt (FETCH_SIZE x 2
            # per the gfx950 correction + WRITE_SIZE); only known for the default workload, null otherwise
            traffic = PMC_TRAFFIC.get((args.material, args.batch, args.precision, "edge_encode"))
            roof = {"bound": "mfma", "kernel": "edge_encode_kernel", "achieved": achieved, "peak": peak,
                    "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                    "traffic_unit": "HBM bytes per launch (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_final_traffic_1stream.txt)",
                    "algorithmic_bytes": e_per * ((320 if args.precision == "fast" else 640) + 68),
                    "avg_launch_ms": ms[k] / cnt[k], "edges_per_launch": e_per, "flop_per_edge": flop_edge,
                    "fp32_equivalent_tflops": FLOP_PER_EDGE * e_per / avg_s / 1e12,
                    "mfma": "v_mfma_f32_32x32x16_bf16 x3 (hi*hi + hi*lo + lo*hi)" if b3 else "v_mfma_f32_32x32x2_f32",
                    "measured": f"HIP events on the launch stream, {n_prof} single-stream passes after the timed region"}
            ka = _lib.KERNEL_CLASSES.index("aggregate")
            if cnt[ka] > 0 and ms[ka] > 0:
                # segment reduce: Eterm streamed once (640 B/edge fp32, 320 B f16), Hs rows gathered (first touch from
                # HBM once per node, then L2), Hr read + agg written per node (SURVEY §8d B_alg terms)
                n_nodes = args.batch * (wl["n_obj"] + synth.MATERIALS[args.material]["n_tools"])
                nbytes = e_per * (320 if args.precision == "fast" else 640) + n_nodes * 3 * 640
                a_s = ms[ka] / cnt[ka] * 1e-3
                roof_hbm = {"bound": "hbm", "kernel": "aggregate_kernel", "achieved": nbytes / a_s / 1e9, "peak": PEAK_HBM_GBS,
                            "unit": "GB/s", "frac": nbytes / a_s / 1e9 / PEAK_HBM_GBS,
                            "traffic": PMC_TRAFFIC.get((args.material, args.batch, args.precision, "aggregate")),
                            "avg_launch_ms": ms[ka] / cnt[ka], "bytes_per_launch": nbytes}

    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        line = {
            "metric": "rollout graph-steps/s (batch x rollout steps / wall; edge build + GNN forward + state update per graph-step)",
            "value": B_global * T * args.steps / dt, "unit": "graph-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.precision == "f32" else "bf16x3", "data": "synthetic",
            "config": {"workload": f"{args.material} {wl['n_obj']}+tool particles, batch {args.batch}/GPU, "
                                   f"{T}-step rollout (BASELINE configs[1])" if args.material == "rope" else
                                   f"{args.material} {wl['n_obj']} particles, batch {args.batch}/GPU, {T}-step rollout",
                       "global_batch": B_global, "rollout_steps": T, "parallelism": f"batch-shard x{world} + all-gather", "rollout_streams": args.streams,
                       "weights": "seed-0 random init (reference default init)", "precision": args.precision,
                       "arithmetic": DTYPE[args.precision]},
            "roofline": roof, "roofline_hbm": roof_hbm if not args.no_profile else None, "kernels": kernels,
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(weights, args.material, wl["n_obj"], wl["kw"], T)
            line["gpu_over_cpu"] = line["value"] / line["cpu_baseline"]["value"]
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
