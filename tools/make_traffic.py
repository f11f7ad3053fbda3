"""HBM bytes per render-kernel launch from the two TCC PMC passes (FETCH_SIZE, WRITE_SIZE; separate
rocprofv3 --pmc runs), with the gfx950 correction MI355X_MICROARCH.md prescribes (FETCH_SIZE x2;
both counters are in KiB).  usage: make_traffic.py <fetch_dir> <write_dir> <out.json> <tag> [config P]"""
import collections
import csv
import glob
import json
import sys

fetch_dir, write_dir, out, tag = sys.argv[1:5]
config = sys.argv[5] if len(sys.argv) > 5 else "nuscenes_gs25600_solid"
P = int(sys.argv[6]) if len(sys.argv) > 6 else 25601


def mean_counter(d, counter, kernel="gf_splat_render"):   # the render kernel of the pass: matrix-core (default flags) or exact-fp32 tile kernel
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and kernel in r["Kernel_Name"]:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    name, vals = max(acc.items(), key=lambda kv: len(kv[1]))
    return name, sum(vals) / len(vals), len(vals)


name, fetch, n = mean_counter(fetch_dir, "FETCH_SIZE")
_, write, _ = mean_counter(write_dir, "WRITE_SIZE")
_, pfetch, _ = mean_counter(fetch_dir, "FETCH_SIZE", "gf_splat_prep_kernel")
_, pwrite, _ = mean_counter(write_dir, "WRITE_SIZE", "gf_splat_prep_kernel")
N = 640000
json.dump({
    "source": f"profiles/pmc_{tag}.txt (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, {n} launches each, {config})",
    "config": config,
    "render_kernel": name[:70],
    "FETCH_SIZE_KiB_raw": fetch,
    "WRITE_SIZE_KiB_raw": write,
    "correction": "gfx950: FETCH_SIZE x2 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported",
    "render_kernel_hbm_bytes_per_launch": int(round((2 * fetch + write) * 1024)),
    "render_kernel_hbm_bytes_per_launch_uncorrected": int(round((fetch + write) * 1024)),
    "prep_kernel_hbm_bytes_per_launch": int(round((2 * pfetch + pwrite) * 1024)),
    "step_hbm_bytes": int(round((2 * (fetch + pfetch) + write + pwrite) * 1024)),
    "algorithmic_bytes_per_launch": 128 * P + 24 * N + 72 * N,
}, open(out, "w"), indent=2)
print(open(out).read())
