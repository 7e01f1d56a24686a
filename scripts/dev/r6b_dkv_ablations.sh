#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6b
for lib in hip abl1 abl2 abl4 abl8 abl12 abl14 abl15 lock hip; do
  echo "## $lib" >> gpurun_out/r6b/dkv_ablations.txt
  CSN_LIB_PATH=csn_amd/libcsn_$lib.so timeout -k 10 200 python scripts/bench_attn.py --tiles --mode 2 --d 96 --nb 100 --evals 80 --slots 40 --recompute 2 --only dkv 2>&1 | grep -v amdgpu >> gpurun_out/r6b/dkv_ablations.txt || exit 1
done
timeout -k 10 300 python scripts/ab_two_streams.py --config 3 --math bf16x3 > gpurun_out/r6b/two_streams_c3.txt 2>&1 || exit 1
timeout -k 10 300 python scripts/ab_two_streams.py --config 5 --math fp16 > gpurun_out/r6b/two_streams_c5.txt 2>&1 || exit 1
timeout -k 10 300 python scripts/ab_two_streams.py --config 3 --math bf16 > gpurun_out/r6b/two_streams_c3_bf16.txt 2>&1
