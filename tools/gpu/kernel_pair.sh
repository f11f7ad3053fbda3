#!/bin/bash
# per-config GPU time of the prep launch and of the render launch (rocprofv3 kernel trace of tools/mfma_probe.py), exact-fp32 and matrix-core paths
export TMPDIR=/tmp; mkdir -p gpurun_out
for cfg in nuscenes_gs25600_solid nuscenes_gs144000; do
rm -rf gpurun_out/kt_$cfg; rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt_$cfg -- python tools/mfma_probe.py $cfg > gpurun_out/kt_$cfg.log 2>&1
python - $cfg <<'PY'
import csv,glob,sys,collections
cfg=sys.argv[1]
f=glob.glob('gpurun_out/kt_%s/**/*kernel_trace.csv'%cfg,recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'gf::' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
acc=collections.defaultdict(list)
for a,b in zip(rows[:-1],rows[1:]):
    if 'prep_kernel' in a['Kernel_Name'] and 'render' in b['Kernel_Name']:
        kind='mfma' if 'mfma' in b['Kernel_Name'] else 'exact'
        acc[kind+' prep'].append(int(a['End_Timestamp'])-int(a['Start_Timestamp']))
        acc[kind+' render'].append(int(b['End_Timestamp'])-int(b['Start_Timestamp']))
        acc[kind+' gap'].append(int(b['Start_Timestamp'])-int(a['End_Timestamp']))
for k,v in sorted(acc.items()):
    v=v[len(v)//4:]
    print(cfg,k,'n=%d mean %.2f us'%(len(v),sum(v)/len(v)/1e3))
PY
grep "us per step" gpurun_out/kt_$cfg.log
done
