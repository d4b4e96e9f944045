#!/bin/bash
# Submit a GPU-box script through gpurun, waiting for a free slot: exit code 3 (no slot, nothing charged) is retried every two
# minutes, any other outcome ends the loop.  Usage: tools/gpu/submit.sh <timeout s> <script> [log]
T=$1; S=$2; L=${3:-gpurun_out/submit_$(basename $S .sh).log}
mkdir -p gpurun_out
for i in $(seq 1 60); do
  /usr/local/graft/bin/gpurun --timeout $T -- "bash $S" > $L 2>&1
  rc=$?
  [ $rc -eq 3 ] || break
  sleep 120
done
echo "submit rc=$rc" >> $L
exit $rc
