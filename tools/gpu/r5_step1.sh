#!/bin/bash
# Run ON THE GPU BOX (via gpurun): the round's new tests, the host-API bench, the default bench line.
mkdir -p gpurun_out
python -m pytest tests/test_hip_screen_bounds.py tests/test_hip_bench_shape.py tests/test_bench_launch.py \
  "tests/test_hip_parity.py::test_sample_in_blocks_equals_one_launch" "tests/test_hip_parity.py::test_sample_plans_blocks_for_a_large_array" \
  "tests/test_hip_parity.py::test_sample_api_semantics" -m gpu -x -q -s > gpurun_out/r5_t1.log 2>&1
rc=$?; echo "rc=$rc" >> gpurun_out/r5_t1.log; tail -30 gpurun_out/r5_t1.log
[ $rc -eq 0 ] || exit $rc
python tools/bench_host_api.py --json gpurun_out/r5_host_api.json > gpurun_out/r5_host_api.log 2>&1 || { tail -20 gpurun_out/r5_host_api.log; exit 1; }
cat gpurun_out/r5_host_api.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r5_bench1.json 2> gpurun_out/r5_bench1.err || { tail -20 gpurun_out/r5_bench1.err; exit 1; }
tail -c 300 gpurun_out/r5_bench1.json; wc -c gpurun_out/r5_bench1.json
