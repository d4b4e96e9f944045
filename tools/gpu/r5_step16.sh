#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_hip_statistics.py tests/test_hip_parity.py -m gpu -q --maxfail=5 -k "statistics or sample or stats" > gpurun_out/r5_t16.log 2>&1
echo "rc=$?"; tail -5 gpurun_out/r5_t16.log | cut -c1-200
