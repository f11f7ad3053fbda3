"""HBM / L2 traffic per launch of the deformable-aggregation and splat-backward kernels from the PMC passes of
tools/gpu/pmc_daf.sh (FETCH_SIZE x2 per MI355X_MICROARCH.md's gfx950 correction, WRITE_SIZE as reported, both in KiB;
TCP_TCC_READ_REQ = L1 -> L2 read requests, 64 B each as counted on gfx950; TCC_HIT / TCC_MISS = L2 hit rate).
usage: make_traffic_daf.py <profiles dir> <tag>   (reads gpurun_out/pmcdaf_<dist>_<pass>/ and gpurun_out/pmcbwd_<pass>/)"""
import collections
import csv
import glob
import json
import sys

out_dir, tag = sys.argv[1:3]


def counters(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0][:64]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


def merge(prefix, passes):
    kern = collections.defaultdict(dict)
    for p in passes:
        for k, cs in counters(f"gpurun_out/{prefix}{p}").items():
            kern[k].update(cs)
    return kern


def summarise(kern, keep):
    rows = {}
    for k, c in kern.items():
        if not any(s in k for s in keep):
            continue
        row = {}
        if "FETCH_SIZE" in c:
            row["hbm_read_bytes"] = int(2 * c["FETCH_SIZE"] * 1024)
        if "WRITE_SIZE" in c:
            row["hbm_write_bytes"] = int(c["WRITE_SIZE"] * 1024)
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
            row["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
            row["l2_requests"] = int(c.get("TCC_REQ_sum", c["TCC_HIT_sum"] + c["TCC_MISS_sum"]))
        if "TCP_TCC_READ_REQ_sum" in c:
            row["l1_to_l2_read_requests"] = int(c["TCP_TCC_READ_REQ_sum"])
            row["l1_to_l2_read_bytes_at_64B"] = int(c["TCP_TCC_READ_REQ_sum"] * 64)
        if "TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum" in c:
            row["l2_atomics_without_return"] = int(c["TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum"])
        if "TCP_TOTAL_CACHE_ACCESSES_sum" in c:
            row["l1_cache_accesses"] = int(c["TCP_TOTAL_CACHE_ACCESSES_sum"])
        if "SQ_WAVE_CYCLES" in c:
            row["sq_wave_cycles"] = int(c["SQ_WAVE_CYCLES"])
            row["sq_wait_any_frac"] = c.get("SQ_WAIT_ANY", 0) / max(c["SQ_WAVE_CYCLES"], 1)
            row["sq_active_valu_frac"] = c.get("SQ_ACTIVE_INST_VALU", 0) / max(c["SQ_WAVE_CYCLES"], 1)
        rows[k] = row
    return rows


res = {"source": f"rocprofv3 --pmc passes of tools/gpu/pmc_daf.sh ({tag}); FETCH_SIZE x2 (gfx950 correction), WRITE_SIZE as reported",
       "shape": "230 400 sample points, 6 cameras, 4 levels (28 700 pixels), 128 channels, 4 groups; algorithmic bytes fwd 305.7 MB"}
for dist in ("uniform", "projected"):
    res["daf_" + dist] = summarise(merge(f"pmcdaf_{dist}_", "CDEFA"), ("gf_daf",))
res["splat_bwd_gs25600"] = summarise(merge("pmcbwd_", "CDEFA"), ("gf_splat_bwd", "gf_bwd"))
print(json.dumps(res, indent=1))
