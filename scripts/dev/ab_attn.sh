#!/bin/bash
# development aid (GPU box): the attention launches alone, several library builds side by side:  ab_attn.sh <lib.so>...
cd "$GRAFT_REPO_ROOT" || exit 1
for g in "--mode 2 --d 96 --nb 100 --evals 80 --slots 40" "--mode 1 --d 96 --nb 100 --evals 80 --slots 40" "--mode 2 --d 128 --evals 256" "--mode 1 --d 128 --evals 256" "--mode 1 --d 64 --evals 256" "--mode 2 --d 64 --evals 256"; do
  for lib in "$@"; do
    echo "== $lib  $g"
    CSN_LIB_PATH=$lib timeout -k 10 120 python scripts/bench_attn.py --tiles $g --only fwd,dq --noscores 2>&1 | grep "^mode" || exit 3
    CSN_LIB_PATH=$lib timeout -k 10 120 python scripts/bench_attn.py --tiles $g --only dq 2>&1 | grep "^mode" || exit 3
    CSN_LIB_PATH=$lib timeout -k 10 120 python scripts/bench_attn.py --tiles $g --recompute 2 --only dq,dkv --noscores 2>&1 | grep "^mode" || exit 3
  done
done
