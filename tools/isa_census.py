"""Static census of a kernel's ISA by source line: compile splat_fwd.hip (or another source) for gfx950 with line tables
(-gline-tables-only keeps the product build's -Os code generation) and count, per source line of the .hip file, the VALU / transcendental / fp64 /
MFMA / SALU / LDS / vector-memory instructions the kernel's code holds for it.  Lines are then grouped into the PHASES given on the
command line (name=first-last, in source lines) and weighted by trip counts (name*count) to give instructions per unit of work.

    python tools/isa_census.py <kernel-name-substring> [--src gaussianformer_amd/csrc/splat_fwd.hip] [--phases file.json]

No GPU needed (hipcc cross-compiles).  The instruction classes: `trans` = v_exp/v_log/v_rcp/v_rsq/v_sqrt/v_sin/v_cos (quarter rate:
16 cycles per wave instruction), `f64` = v_*_f64 (incl. conversions), `valu` = every other v_* that is not an MFMA, `dpp` = the
subset of valu carrying a DPP / SDWA modifier (cross-lane)."""
import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith("v_"):
        if op.startswith(TRANS):
            return "trans"
        if "_f64" in op:
            return "f64"
        return "valu"
    if op.startswith("s_"):
        if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep", "s_setprio")):
            return "wait"
        if op.startswith(("s_load", "s_buffer_load")):
            return "smem"
        if op.startswith(("s_cbranch", "s_branch")):
            return "branch"
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def assemble(src, extra):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = ["hipcc", "--offload-arch=gfx950", "-Os", "-std=c++17", "-fPIC", "-Wno-inline-asm", "-gline-tables-only",
           "--cuda-device-only", "-S", src, "-o", out] + extra
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return out


def census(asm_path, kernel_sub, src_base):
    lines = open(asm_path).read().split("\n")
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m:
            files[int(m.group(1))] = m.group(3) or m.group(2)
    start = end = None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m and kernel_sub in m.group(1) and start is None:
            start = i
        if start is not None and end is None and re.match(r"^\.Lfunc_end\d+:", l):
            end = i
    assert start is not None, f"no kernel matching {kernel_sub}"
    per_line = collections.defaultdict(collections.Counter)
    cur = (None, 0)
    for l in lines[start:end]:
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
            # inlined code: attribute to the OUTERMOST frame (the kernel's own line), from the "; file:line:col @[ caller ... ]" comment
            frames = re.findall(r"([^\s\[\]]+):(\d+):\d+", l.split(";", 1)[1]) if ";" in l else []
            if frames:
                cur = (frames[-1][0], int(frames[-1][1]))
            continue
        if not l.startswith("\t"):
            continue
        t = l.strip()
        if not t or t[0] in ".;":
            continue
        op = t.split()[0]
        c = classify(op)
        key = cur[1] if os.path.basename(cur[0] or "") == src_base else ("other:" + os.path.basename(cur[0] or "?"), 0)
        per_line[key][c] += 1
        if c == "valu" and ("dpp" in t or "row_" in t or "quad_perm" in t):
            per_line[key]["dpp"] += 1
    return per_line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("kernel")
    ap.add_argument("--src", default=os.path.join(ROOT, "gaussianformer_amd", "csrc", "splat_fwd.hip"))
    ap.add_argument("--phases", default=None, help="JSON: [[name, first_line, last_line, trips_per_unit], ...]")
    ap.add_argument("--flags", default="", help="extra hipcc flags")
    ap.add_argument("--lines", action="store_true", help="print the per-line table too")
    a = ap.parse_args()
    asm = assemble(a.src, a.flags.split())
    per_line = census(asm, a.kernel, os.path.basename(a.src))
    os.unlink(asm)
    classes = ["valu", "trans", "f64", "mfma", "salu", "branch", "wait", "smem", "lds", "vmem", "dpp"]
    tot = collections.Counter()
    for c in per_line.values():
        tot.update(c)
    print("static totals:", {k: tot[k] for k in classes})
    if a.lines:
        for k in sorted((k for k in per_line if isinstance(k, int))):
            print(k, dict(per_line[k]))
        for k in (k for k in per_line if not isinstance(k, int)):
            print(k, dict(per_line[k]))
    if a.phases:
        phases = json.load(open(a.phases))
        used = set()
        print(f"{'phase':44s} {'trips':>6s} | " + " ".join(f"{c:>6s}" for c in classes) + " | per unit: valu+trans+f64   cycles(4/16/8)")
        gsum = collections.Counter()
        gcyc = 0.0
        for name, lo, hi, trips in phases:
            c = collections.Counter()
            for k, v in per_line.items():
                if isinstance(k, int) and lo <= k <= hi:
                    c.update(v)
                    used.add(k)
            v_all = c["valu"] + c["trans"] + c["f64"]
            cyc = 4 * c["valu"] + 16 * c["trans"] + 8 * c["f64"]
            print(f"{name:44s} {trips:6.2f} | " + " ".join(f"{c[x]:6d}" for x in classes) + f" | {v_all * trips:8.1f} {cyc * trips:10.0f}")
            for x in classes:
                gsum[x] += c[x] * trips
            gcyc += cyc * trips
        rest = collections.Counter()
        for k, v in per_line.items():
            if k not in used:
                rest.update(v)
        print(f"{'(not in any phase)':44s} {'':6s} | " + " ".join(f"{rest[x]:6d}" for x in classes))
        print("per unit:", {x: round(gsum[x], 1) for x in classes}, "VALU issue cycles per unit", round(gcyc))


if __name__ == "__main__":
    main()
