#!/bin/bash
# development aid (GPU box, via gpurun): parity tests of the score-recomputing flows, then kernel-level A/B timings of the
# attention launches in each flow:  ab_flows.sh <tag>   ->  gpurun_out/<tag>_{tests,ab}.log
cd "$GRAFT_REPO_ROOT" || exit 1
T=gpurun_out/$1
timeout -k 10 420 python -m pytest tests/test_gpu_flash.py -x -q > ${T}_tests.log 2>&1
rc=$?
tail -5 ${T}_tests.log
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "tests ended with $rc: not timing"; exit $rc; fi
{
for geo in "--mode 2 --evals 256" "--mode 2 --d 96 --nb 100 --evals 80 --slots 40" "--mode 1 --d 96 --nb 100 --evals 80 --slots 40" "--mode 2 --d 128 --evals 256" "--mode 1 --d 128 --evals 256"; do
  timeout -k 10 120 python scripts/bench_attn.py --tiles $geo --only fwd,dq,dkv || exit 3
  timeout -k 10 120 python scripts/bench_attn.py --tiles $geo --only fwd --noscores || exit 3
  timeout -k 10 120 python scripts/bench_attn.py --tiles $geo --recompute 1 --only dq || exit 3
  timeout -k 10 120 python scripts/bench_attn.py --tiles $geo --recompute 2 --only dq,dkv || exit 3
done
timeout -k 10 120 python scripts/bench_attn.py --tiles --mode 1 --evals 256 --only fwd,dq,dkv || exit 3
} > ${T}_ab.log 2>&1
tail -30 ${T}_ab.log
exit $rc
