#!/usr/bin/env python3
"""GPU box: collect per-kernel HBM traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, SEPARATE passes, no trace domains)
for the bench workloads and write profiles/pmc_traffic.json, keyed by the sha256 of the kernel sources so that bench.py
only quotes it while those sources are the ones running.

    python tools/pmc_traffic.py [--out gpurun_out/pmc_traffic.json]     (copy the result to profiles/)

Corrections per MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports half the bytes of wide streaming reads,
so bytes = 2 x FETCH_SIZE(KB) x 1024 + WRITE_SIZE(KB) x 1024, per dispatch; dispatches are full-batch launches
(--streams 1)."""
import argparse, json, os, re, sqlite3, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (csrc_sha)

RUNS = [("rope", 256, 10, "fast"), ("rope", 256, 10, "f32"), ("granular", 128, 10, "fast"), ("cloth", 64, 20, "fast")]
KERNELS = {"edge_encode": ("edge_encode_kernel", "edge_encode_nb_kernel", "edge_encode_ws_kernel"), "aggregate": ("aggregate_half_kernel", "aggregate_kernel"),
           "node_update": ("node_update_kernel", "node_update_nws_kernel"), "node_update_stationary": ("node_update_nws_kernel",),
           "node_update_last": ("node_update_kernel",), "node_encode": ("node_encode_kernel",)}


def one_pass(counter, mat, batch, T, prec, tmp):
    out = os.path.join(tmp, f"{counter}_{mat}_{prec}")
    cmd = ["rocprofv3", "--pmc", counter, "-d", out, "-o", "pmc", "--", sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1",
           "--warmup", "1", "--no-cpu-baseline", "--no-profile", "--no-extra", "--streams", "1", "--material", mat, "--batch", str(batch),
           "--rollout-steps", str(T), "--precision", prec]
    subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    db = [os.path.join(dp, f) for dp, _, fs in os.walk(out) for f in fs if f.endswith(".db")][0]
    acc = {}
    for kn, v in sqlite3.connect(db).execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
        a = acc.setdefault(kn, [0.0, 0])
        a[0] += v; a[1] += 1
    return acc          # kernel name -> [sum over dispatches, dispatches]


def per_dispatch(acc, names):
    """Average over all DISPATCHES of the kernels whose name contains one of `names` (a class with two kernels is weighted by launches)."""
    hit = [v for k, v in acc.items() if any(re.search(r"\b" + n + r"\b", k) for n in names)]
    n = sum(c for _, c in hit)
    return sum(s for s, _ in hit) / n if n else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "pmc_traffic.json"))
    a = ap.parse_args()
    entries, raw = {}, {}
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        for mat, batch, T, prec in RUNS:
            f = one_pass("FETCH_SIZE", mat, batch, T, prec, tmp)
            w = one_pass("WRITE_SIZE", mat, batch, T, prec, tmp)
            for key, names in KERNELS.items():
                fk, wk = per_dispatch(f, names), per_dispatch(w, names)
                if fk is None:
                    continue
                entries[f"{mat}/{batch}/{prec}/{key}"] = (2 * fk + wk) * 1024
                raw[f"{mat}/{batch}/{prec}/{key}"] = {"FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk}
    rec = {"csrc_sha256": bench.csrc_sha(), "entries": entries, "raw_kb_per_dispatch": raw,
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --steps 1 --warmup 1 --streams 1; bytes = 2 x FETCH + WRITE (gfx950 correction)"}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(rec, open(a.out, "w"), indent=1)
    print(json.dumps(entries, indent=1))


if __name__ == "__main__":
    main()
