#!/bin/bash
# development aid: three PMC passes over a python script (run on the GPU box via gpurun):  run_pmc.sh <outdir> <kernel-filter> <script> [args]
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; FLT=$2; shift; shift
mkdir -p $OUT
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES -d $OUT/a -- python3 "$@" > $OUT/a.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS -d $OUT/b -- python3 "$@" > $OUT/b.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d $OUT/c -- python3 "$@" > $OUT/c.log 2>&1
python3 scripts/pmc_summary.py $OUT "$FLT" > $OUT/summary.txt 2>&1
rm -rf $OUT/a $OUT/b $OUT/c          # the raw traces are large: only the summary travels back
