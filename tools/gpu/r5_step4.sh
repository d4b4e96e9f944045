#!/bin/bash
# Run ON THE GPU BOX: A/B of the service-phase thresholds of the lane kernels after the move to four tries per attempt.
mkdir -p gpurun_out
tools/ab_libs.sh "libgsss_hip.so libgsss_w12.so libgsss_w23.so libgsss_w78.so libgsss_p34.so libgsss_hip.so" "vmfmix_readme:1000000 vmfmix_k10_kappa500:1000000" > gpurun_out/r5_ab_service.log 2>&1
cat gpurun_out/r5_ab_service.log
