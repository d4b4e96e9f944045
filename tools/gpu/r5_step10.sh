#!/bin/bash
# Run ON THE GPU BOX: the driver's bench command, timed.
mkdir -p gpurun_out
SECONDS=0
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_bench_driver.json 2> gpurun_out/r5_bench_driver.err
echo "rc=$? wall seconds: $SECONDS"
tail -c 1200 gpurun_out/r5_bench_driver.json; echo; wc -c gpurun_out/r5_bench_driver.json
cp bench_full.json gpurun_out/r5_bench_driver_full.json
