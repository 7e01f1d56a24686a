#!/bin/bash
# round 6 final evidence (GPU box): bench lines of every configuration / mode, SQ counter passes of the attention launches at
# configs 3 and 5, the published geometry (n_heads = 8, K = 4) with the counters of its out-projection launch
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1
bash scripts/dev/bench_lines.sh $tag || exit 3
timeout -k 10 300 python bench.py --heads 8 --K 4 --shapes 8 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_h8k4.json || exit 3
bash scripts/run_pmc.sh ${tag}_sq_c3_bf16x3 csn_attn bench.py --steps 2 --warmup 1 --headline-only --no-cpu-baseline || exit 4
bash scripts/run_pmc.sh ${tag}_sq_c5_fp16 csn_attn bench.py --config 5 --math fp16 --steps 2 --warmup 1 --headline-only --no-cpu-baseline || exit 4
bash scripts/run_pmc.sh ${tag}_sq_c5_bf16x3 csn_attn bench.py --config 5 --steps 2 --warmup 1 --headline-only --no-cpu-baseline || exit 4
export TMPDIR=/tmp
O=gpurun_out/${tag}_h8k4; mkdir -p $O
B="bench.py --heads 8 --K 4 --shapes 8 --steps 1 --warmup 1 --headline-only --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/p1 -- python3 $B > $O/p1.log 2>&1 || exit 5
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/p2 -- python3 $B > $O/p2.log 2>&1 || exit 5
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $O/p3 -- python3 $B > $O/p3.log 2>&1 || exit 5
{ echo "# python3 $B: the out-projection + LayerNorm launch at K = n_heads x d_v = 2048 (tiled kernel) and the attention launches";
  python3 scripts/pmc_summary.py $O/p1 csn_gemm_bf16x3_big; python3 scripts/pmc_summary.py $O/p2 csn_gemm_bf16x3_big; python3 scripts/pmc_summary.py $O/p3 csn_gemm_bf16x3_big; } > $O/outproj_counters.txt
rm -rf $O/p1 $O/p2 $O/p3
cat $O/outproj_counters.txt
