#!/bin/bash
# development aid: PMC passes over the attention microbenchmark (run on the GPU box via gpurun)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
ARGS="$@"
mkdir -p $OUT
timeout -k 10 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES -d $OUT/a -- python3 scripts/bench_attn.py $ARGS > $OUT/a.log 2>&1 &&
timeout -k 10 150 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS -d $OUT/b -- python3 scripts/bench_attn.py $ARGS > $OUT/b.log 2>&1 &&
timeout -k 10 150 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d $OUT/c -- python3 scripts/bench_attn.py $ARGS > $OUT/c.log 2>&1
python3 scripts/pmc_summary.py $OUT csn_attn > $OUT/summary.txt 2>&1
