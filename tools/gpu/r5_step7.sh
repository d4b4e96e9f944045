#!/bin/bash
# Run ON THE GPU BOX: parity of the curve kernels, then the knot-row pipeline in the three-quad three-wavefront builds on (libgsss_pipeq3.so) / off (default).
mkdir -p gpurun_out
python -m pytest tests/test_hip_parity.py tests/test_hip_bench_shape.py tests/test_hip_fuzz.py -m gpu --maxfail=5 -q -k "curve or synthetic" > gpurun_out/r5_t7.log 2>&1
rc=$?; echo "rc=$rc" >> gpurun_out/r5_t7.log; tail -5 gpurun_out/r5_t7.log
[ $rc -eq 0 ] || exit $rc
tools/ab_libs.sh "libgsss_hip.so libgsss_pipeq3.so libgsss_hip.so libgsss_pipeq3.so" "curve_d40:100000 curve_d50:100000 curve_d80:100000 curve_d100:100000 curve_d160:100000 curve_d200:100000" > gpurun_out/r5_ab_pipeq3.log 2>&1
cat gpurun_out/r5_ab_pipeq3.log
export TMPDIR=/tmp
for LIB in libgsss_hip.so libgsss_pipeq3.so; do
for W in curve_d50 curve_d200; do
  for P in FETCH_SIZE WRITE_SIZE; do
    GSSS_HIP_LIB=$PWD/geosss_amd/$LIB rocprofv3 --pmc $P --output-format csv -d gpurun_out/qt_${LIB}_${W}_$P -- python3 bench.py --workload $W --chains 100000 --steps 3 --warmup 1 --no-configs --no-cpu-baseline --no-ess > gpurun_out/qt_${LIB}_${W}_$P.log 2>&1 || exit 1
  done
  echo "$LIB $W"; python3 tools/pmc_summary.py gpurun_out/qt_${LIB}_${W}_FETCH_SIZE gpurun_out/qt_${LIB}_${W}_WRITE_SIZE | grep -A1 curvespec | grep SIZE
done
done
