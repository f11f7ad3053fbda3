#!/bin/bash
export TMPDIR=/tmp
for i in 1 2; do
timeout 200 python tools/bwd_probe.py full nuscenes_gs25600_solid 2>&1 | grep "forward " | sed 's/^/build lists: /' | cut -c1-150
GF_FWD_CONSUME_LISTS=1 timeout 200 python tools/bwd_probe.py full nuscenes_gs25600_solid 2>&1 | grep "forward \|vs oracle" | sed 's/^/consume:     /' | cut -c1-150
done
