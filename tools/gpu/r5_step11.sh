#!/bin/bash
mkdir -p gpurun_out
for f in examples/*.py; do
  SECONDS=0
  timeout -k 10 300 python $f > gpurun_out/r5_example_$(basename $f .py).log 2>&1
  echo "$f rc=$? ${SECONDS}s"; tail -3 gpurun_out/r5_example_$(basename $f .py).log | cut -c1-200
done
