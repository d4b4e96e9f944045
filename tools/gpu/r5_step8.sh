#!/bin/bash
# Run ON THE GPU BOX: the README kernel two chains per lane (default) against the one-chain-per-lane BUILD (-DGSSS_VMF_ONE_ALL=1,
# GSSS_ONE_PER_LANE=2) after the move to four tries per attempt; smoke().
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5_smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/r5_smoke.log
for i in 1 2; do
tools/ab_libs.sh "libgsss_hip.so" "vmfmix_readme:1000000"
GSSS_ONE_PER_LANE=2 tools/ab_libs.sh "libgsss_oneall.so" "vmfmix_readme:1000000"
GSSS_ONE_PER_LANE=2 tools/ab_libs.sh "libgsss_hip.so" "vmfmix_readme:1000000"
done
