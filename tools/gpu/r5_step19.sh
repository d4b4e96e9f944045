#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_hip_parity.py tests/test_integration_binding.py -m gpu -q --maxfail=5 -k "sample or binding" > gpurun_out/r5_t19.log 2>&1
echo "rc=$?"; tail -5 gpurun_out/r5_t19.log | cut -c1-200
python tools/bench_host_api.py --json gpurun_out/r5_host_api.json > gpurun_out/r5_host_api.log 2>&1; cat gpurun_out/r5_host_api.log | cut -c1-400
