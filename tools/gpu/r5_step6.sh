#!/bin/bash
# Run ON THE GPU BOX: knot-row pipeline on / off in the three-wavefront curve builds (scratch 240 -> 152 B a lane at d = 200).
mkdir -p gpurun_out
tools/ab_libs.sh "libgsss_hip.so libgsss_nopipe.so libgsss_hip.so libgsss_nopipe.so" "curve_d24:100000 curve_d40:100000 curve_d50:100000 curve_d80:100000 curve_d100:100000 curve_d160:100000 curve_d200:100000" > gpurun_out/r5_ab_nopipe.log 2>&1
cat gpurun_out/r5_ab_nopipe.log
export TMPDIR=/tmp
for W in curve_d50 curve_d200; do
  for P in FETCH_SIZE WRITE_SIZE; do
    GSSS_HIP_LIB=$PWD/geosss_amd/libgsss_nopipe.so rocprofv3 --pmc $P --output-format csv -d gpurun_out/qt_nopipe_${W}_$P -- python3 bench.py --workload $W --chains 100000 --steps 3 --warmup 1 --no-configs --no-cpu-baseline --no-ess > gpurun_out/qt_nopipe_${W}_$P.log 2>&1 || exit 1
  done
  python3 tools/pmc_summary.py gpurun_out/qt_nopipe_${W}_FETCH_SIZE gpurun_out/qt_nopipe_${W}_WRITE_SIZE | grep -A1 curvespec
done
