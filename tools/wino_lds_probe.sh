#!/bin/bash
# Where do the Winograd kernel's LDS bank conflicts come from?  Probe builds (-DCLSLAM_WINO_DBG: 16 no input transform, 4 no MFMA
# (= no U operand reads), 8 no DMA) of the library ON THE GPU BOX, one --pmc pass each over tools/bench_conv.py on a 128-channel layer:
#   bash tools/wino_lds_probe.sh > gpurun_out/r05_wino_lds_probe.txt     (restores the production library at the end)
export TMPDIR=/tmp
for dbg in ${PROBE_DBG:-0 16 4 20}; do
    CLSLAM_HIPCC_EXTRA="-DCLSLAM_WINO_DBG=$dbg" python cl-slam_amd/csrc/build.py --force > /tmp/build_$dbg.log 2>&1
    rm -rf /tmp/pmc_w
    (cd /tmp && CLSLAM_HIPCC_EXTRA="-DCLSLAM_WINO_DBG=$dbg" BENCH_WGRAD=0 BENCH_LAYERS=1 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/pmc_w -o run -- python $OLDPWD/tools/bench_conv.py 10 40 > /tmp/pmc_w.log 2>&1)
    echo "== CLSLAM_WINO_DBG=$dbg  (`grep layer2 /tmp/pmc_w.log | cut -c1-90`)"
    python tools/pmc_summary.py $(ls /tmp/pmc_w/*/*.db /tmp/pmc_w/*.db 2>/dev/null | head -1) wino8 2>&1 | tail -3
done
python cl-slam_amd/csrc/build.py --force > /tmp/build_prod.log 2>&1
