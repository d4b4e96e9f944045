#!/bin/bash
mkdir -p gpurun_out
GSSS_FUZZ_SCALE=10 python -m pytest tests/test_hip_fuzz.py -m gpu -q --maxfail=5 > gpurun_out/r5_fuzz_soak.log 2>&1
echo "rc=$?"; tail -6 gpurun_out/r5_fuzz_soak.log | cut -c1-200
