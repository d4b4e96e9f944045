#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: collects the rocprofv3 evidence bench.py's
# roofline numbers are checked against.  Usage: tools/collect_profiles.sh <tag> [bench args...]
# Kernel trace and PMC passes are separate runs (gpurun refuses --pmc together with trace domains).
set -u
TAG=${1:-r01}; shift || true
OUT=gpurun_out/prof_$TAG
export TMPDIR=/tmp
mkdir -p $OUT
ARGS="--steps 5 --warmup 1 --no-cpu-baseline --no-ess $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/trace.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py $ARGS > $OUT/fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py $ARGS > $OUT/write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY \
  --output-format csv -d $OUT/sq1 -- python3 bench.py $ARGS > $OUT/sq1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FLOPS_FP64 \
  --output-format csv -d $OUT/sq2 -- python3 bench.py $ARGS > $OUT/sq2.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE \
  --output-format csv -d $OUT/sq3 -- python3 bench.py $ARGS > $OUT/sq3.log 2>&1 || exit 1
{
  echo "# rocprofv3 summary ($TAG): python3 bench.py $ARGS"
  echo; echo "## bench line (from the kernel-trace run)"; grep -h '^{"metric"' $OUT/trace.log
  echo; echo "## kernel-trace --stats (top kernels)"; head -6 $OUT/trace/*/*_kernel_stats.csv
  echo; echo "## PMC (mean per dispatch)"; python3 tools/pmc_summary.py $OUT/fetch $OUT/write $OUT/sq1 $OUT/sq2 $OUT/sq3
} > $OUT/summary.md
cat $OUT/summary.md | cut -c1-220
