#!/bin/bash
# Run ON THE GPU BOX: the GPU suite twice more in fresh processes (a flaky test would show), one process at a time.
mkdir -p gpurun_out
for i in 1 2; do
  python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > gpurun_out/r5_repeat_$i.log 2>&1
  rc=$?; echo "run $i rc=$rc: $(tail -1 gpurun_out/r5_repeat_$i.log)"
  [ $rc -eq 0 ] || { tail -30 gpurun_out/r5_repeat_$i.log | cut -c1-200; exit $rc; }
done
