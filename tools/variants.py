"""Development aid: build render-kernel variants (-D switches) and time each on the GPU.
usage: python tools/variants.py "name:-DFLAG=1 -DOTHER=2" ...   (name 'base' = no flags)"""
import json
import os
import subprocess
import sys
sys.path.insert(0, ".")
from gaussianformer_amd import build as B
specs = sys.argv[1:] or ["base:"]
for spec in specs:
    name, _, flags = spec.partition(":")
    lib = B.build(extra_flags=tuple(flags.split()), lib_name=f"libgf_hip_{name}.so")
    env = dict(os.environ, GF_LIB=lib)
    print(f"==== variant {name} [{flags}]", flush=True)
    r = subprocess.run([sys.executable, "bench.py", "--steps", "300", "--warmup", "30", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True)
    try:
        j = json.loads(r.stdout.strip().splitlines()[-1])
        print(f"   us/step {j['ms_per_step']*1e3:.1f}  render kernel us {j['roofline']['kernel_us']:.1f}  "
              f"value {j['value']/1e9:.3f} G/s", flush=True)
    except Exception as e:
        print("   FAILED", e, r.stdout[-500:], r.stderr[-1500:], flush=True)
