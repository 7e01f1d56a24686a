#!/bin/bash
# round 6 host-side experiment (b), the upper bound first: the dQ launch and the two plane products per evaluation when the launch
# covers only E evaluations — at E <= 4 the P / dS planes (41 MB per evaluation) of the timed repetitions stay inside the 256 MiB
# Infinity Cache, at E = 64 they cannot.  (scripts/bench_attn.py times each entry point alone, back to back on the same buffers.)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$1; mkdir -p $O
for E in 2 4 8 16 64 256; do
  echo "## evaluations per launch: $E" >> $O/groups.txt
  timeout -k 10 200 python scripts/bench_attn.py --tiles --mode 1 --evals $E --slots $(( E < 32 ? E : 32 )) --only dq,dkv 2>&1 | grep -v amdgpu >> $O/groups.txt || exit 1
done
cat $O/groups.txt
# published geometry (n_heads = 8, K = 4): kernel statistics + SQ counters of the out-projection on the tiled kernel
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --heads 8 --K 4 --shapes 8 --steps 3 --warmup 1 --headline-only --no-cpu-baseline > $O/h8k4_kt.log 2>&1 || exit 1
cp $(ls $O/kt/*/*kernel_stats.csv | head -1) $O/h8k4_kernel_stats.csv; rm -rf $O/kt
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/p1 -- python3 bench.py --heads 8 --K 4 --shapes 8 --steps 1 --warmup 1 --headline-only --no-cpu-baseline > $O/p1.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/p2 -- python3 bench.py --heads 8 --K 4 --shapes 8 --steps 1 --warmup 1 --headline-only --no-cpu-baseline > $O/p2.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $O/p3 -- python3 bench.py --heads 8 --K 4 --shapes 8 --steps 1 --warmup 1 --headline-only --no-cpu-baseline > $O/p3.log 2>&1 || exit 1
{ python3 scripts/pmc_summary.py $O/p1 csn_outproj; python3 scripts/pmc_summary.py $O/p2 csn_outproj; python3 scripts/pmc_summary.py $O/p3 csn_outproj; } > $O/h8k4_outproj_counters.txt
rm -rf $O/p1 $O/p2 $O/p3
cat $O/h8k4_outproj_counters.txt; head -8 $O/h8k4_kernel_stats.csv | cut -c1-160
