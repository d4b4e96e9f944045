#!/bin/bash
# Run ON THE GPU BOX: what the driver runs at round end -- the GPU suite with -x, then smoke().
mkdir -p gpurun_out
python -m pytest tests/ -x -q -m gpu > gpurun_out/r5_fullsuite.log 2>&1
rc=$?; echo "rc=$rc"; tail -6 gpurun_out/r5_fullsuite.log | cut -c1-200
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
