"""Static ISA census of the default forward kernel (gf_splat_render_mfma_wave_kernel<false, false, true>) by phase x trip count
(VERDICT r5 #3c).  Finds the phase boundaries from the kernel's own comment markers, so it survives edits; trip counts per unit are
the headline shape's (nuscenes_gs25600_solid: 2 048 waves for 5 000 units, 140 candidates and 2.16 groups per unit).
    python tools/census_wave.py > profiles/census_wave_r06.txt          (no GPU needed)"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gaussianformer_amd", "csrc", "splat_fwd.hip")
lines = open(SRC).read().split("\n")


def find(marker, after=0):
    for i in range(after, len(lines)):
        if marker in lines[i]:
            return i + 1
    raise SystemExit(f"marker not found: {marker}")


k0 = find("void gf_splat_render_mfma_wave_kernel(RenderArgs a)")
ctr = find("uint32_t *ctr = a.tile_counters + 64 * xcd;", k0)
fast = find("// ---- fill, fast path", k0)
slow = find("while (true) {  // fill the list from the row", k0)
boxes = find("// ---- the packed boxes of the listed Gaussians", k0)
cons = find("// ---- consume: hits of the double brick", k0)
oper = find("// ---- operands of the group: lane", k0)
pair = find("auto pair = [&](int b0) {", k0)
brk = find("if (last) break;", k0)
nxt = find("// ---- the next unit (claimed during the last group)", k0)
epi = find("// ---- accumulators C[channel (q&3)", k0)
end = find("static int mfma_wave_grid", k0)
phases = [
    ["start-up, per wave (2 048 waves / 5 000 units; includes the never-taken arbitrary-points body: 151 VALU)", k0, ctr - 1, 2048 / 5000],
    ["unit head: decode, row request, wait", ctr, fast - 1, 1.0],
    ["list, fast path: dense-word compaction + three-round id extraction", fast, slow - 1, 1.0],
    ["list, chunked refill path (crowded rows only)", slow, boxes - 1, 0.0],
    ["boxes by LDS-DMA (64 candidates per trip)", boxes, cons - 1, 140 / 64 + 0.5],
    ["filter batch (64 candidates per trip)", cons, oper - 1, 140 / 64 + 0.5],
    ["group: operands (theta in fp64 -> 3 x f16, one-hot, S' transpose + split)", oper, pair - 1, 2.16],
    ["group: two pairs of blocks (64 exp, hi/lo split, 40 MFMA)", pair, brk - 1, 2.16],
    ["next claim + next row request", brk, epi - 1, 1.0],
    ["epilogue: stage, ten addresses, ten stores", epi, end - 1, 1.0],
]
with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
    json.dump(phases, f)
print("# tools/census_wave.py -- static instruction counts of gf_splat_render_mfma_wave_kernel<false,false,true> (gfx950, -Os: the product build since the second half of round 6; -O3 before: profiles/census_wave_O3_r06.txt) per source")
print("# phase, weighted by trips per unit at nuscenes_gs25600_solid.  Cycle weights: VALU 4, transcendental 16 (quarter rate), fp64 8")
print("# (upper bound: v_fma_f64 issues at full rate on this part).  PMC of the same kernel (profiles/pmc_wave_vs_solo_r05.txt): 8.39 M VALU")
print("# wave-instructions per launch = 1 680 per unit; SALU 3.86 M = 772 per unit.")
sys.stdout.flush()
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_census.py"), "gf_splat_render_mfma_wave_kernelILb0ELb0ELb1E", "--phases", f.name], check=True)
os.unlink(f.name)
