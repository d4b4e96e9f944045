#!/bin/bash
mkdir -p gpurun_out
python tools/bench_curve_sweep.py 3 6 9 12 15 18 21 24 > gpurun_out/r5_curve_sweep_ref.log 2>&1; cat gpurun_out/r5_curve_sweep_ref.log
python tools/bench_curve_sweep.py 4 10 16 17 32 33 40 48 49 50 52 53 64 65 80 96 97 100 104 105 128 129 160 192 193 200 208 209 256 > gpurun_out/r5_curve_sweep_all.log 2>&1; cat gpurun_out/r5_curve_sweep_all.log
