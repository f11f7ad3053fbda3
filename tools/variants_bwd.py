"""Development aid: build backward-kernel variants (-D switches) and time each on the GPU.
usage: python tools/variants_bwd.py "name:-DFLAG=1" ...   (name 'base' = no flags)"""
import os
import subprocess
import sys
sys.path.insert(0, ".")
from gaussianformer_amd import build as B
for spec in sys.argv[1:] or ["base:"]:
    name, _, flags = spec.partition(":")
    lib = B.build(extra_flags=tuple(flags.split()), lib_name=f"libgf_hip_{name}.so")
    print(f"==== variant {name} [{flags}]", flush=True)
    r = subprocess.run([sys.executable, "tools/bench_ops.py", "--splat-only"], env=dict(os.environ, GF_LIB=lib),
                       capture_output=True, text=True)
    for line in r.stdout.splitlines():
        if "splat_backward" in line:
            print("   ", line[:120], flush=True)
    if r.returncode:
        print("   FAILED", r.stderr[-800:], flush=True)
