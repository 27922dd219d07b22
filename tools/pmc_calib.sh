#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of rocprofv3 against known byte counts in conv_wino.hip's access patterns (tools/micro/pmc_calib.hip):
#   bash tools/pmc_calib.sh > gpurun_out/rNN_pmc_calibration.txt      (on the GPU box)
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -o /tmp/pmc_calib tools/micro/pmc_calib.hip 2>/dev/null || exit 1
for ctr in WRITE_SIZE FETCH_SIZE; do
  rm -rf /tmp/pc_$ctr
  (cd /tmp && rocprofv3 --pmc $ctr -d /tmp/pc_$ctr -o run -- /tmp/pmc_calib > /tmp/pc_$ctr.log 2>&1)
  echo "== $ctr (rocprofv3 reports KB; useful bytes per launch: 67108864 = 65536 KB)"
  python tools/pmc_summary.py $(ls /tmp/pc_$ctr/*/*.db /tmp/pc_$ctr/*.db 2>/dev/null | head -1) "" 2>&1 | grep -v "^$"
done
