#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_hip_parity.py -m gpu -q --maxfail=5 -k "sample" > gpurun_out/r5_t17.log 2>&1
echo "rc=$?"; tail -5 gpurun_out/r5_t17.log | cut -c1-200
