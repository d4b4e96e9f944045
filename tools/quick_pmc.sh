#!/bin/bash
# Run ON THE GPU BOX: the issue-side counters of one workload's dominant kernel (three PMC passes), short form.
# Usage: tools/quick_pmc.sh <tag> <workload> <chains>
set -u
TAG=$1; W=$2; CH=$3
OUT=gpurun_out/qp_$TAG
export TMPDIR=/tmp
mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-ess --workload $W --chains $CH --no-configs"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY \
  --output-format csv -d $OUT/sq1 -- python3 bench.py $ARGS > $OUT/sq1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FLOPS_FP64 \
  --output-format csv -d $OUT/sq2 -- python3 bench.py $ARGS > $OUT/sq2.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE \
  --output-format csv -d $OUT/sq3 -- python3 bench.py $ARGS > $OUT/sq3.log 2>&1 || exit 1
grep -h '^{"metric"' $OUT/sq1.log | python3 -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(j['config']['target'], '%.3e' % j['value'], j['config']['kernel'], '%.2f ms' % j['kernel_ms'])"
python3 tools/pmc_summary.py $OUT/sq1 $OUT/sq2 $OUT/sq3 | grep -A9 "curvespec\|screened\|coopfast\|curve64" | grep -v "^--"
