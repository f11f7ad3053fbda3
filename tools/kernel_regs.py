"""Registers, spills and LDS of the kernels of a HIP source (metadata of the gfx950 code object; no GPU needed).
    python tools/kernel_regs.py [source.hip] [name-substring] [-- extra hipcc flags]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--")
    extra = args[i + 1:]
    args = args[:i]
src = args[0] if args else os.path.join(ROOT, "gaussianformer_amd", "csrc", "splat_fwd.hip")
sub = args[1] if len(args) > 1 else ""
out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
subprocess.run(["hipcc", "--offload-arch=gfx950", "-Os", "-std=c++17", "-fPIC", "-Wno-inline-asm", "--cuda-device-only", "-S", src, "-o", out] + extra,
               check=True, stderr=subprocess.DEVNULL)
t = open(out).read()
os.unlink(out)
for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", t, re.S):
    blk = m.group(0)
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if sub not in name:
        continue
    g = lambda k: (re.search(r"\." + k + r":\s+(\d+)", blk) or [None, "?"])[1]
    print(f"{name[:100]:100s} vgpr {g('vgpr_count')} agpr {g('agpr_count')} sgpr {g('sgpr_count')} sgpr_spill {g('sgpr_spill_count')} "
          f"vgpr_spill {g('vgpr_spill_count')} scratch {g('private_segment_fixed_size')} lds {g('group_segment_fixed_size')}")
