#!/bin/bash
# development aid (GPU box): the d = 256 attention launches (config-3 geometry) of several library builds:  ab_attn256.sh <lib.so>...
cd "$GRAFT_REPO_ROOT" || exit 1
for g in "--mode 2 --evals 256" "--mode 1 --evals 256"; do
  for lib in "$@"; do
    echo "== $lib  $g"
    CSN_LIB_PATH=$lib timeout -k 10 120 python scripts/bench_attn.py --tiles $g --only fwd,dq --noscores 2>&1 | grep "^mode" || exit 3
    CSN_LIB_PATH=$lib timeout -k 10 120 python scripts/bench_attn.py --tiles $g --only fwd,dq 2>&1 | grep "^mode" || exit 3
    if [[ "$g" == *"mode 2"* ]]; then CSN_LIB_PATH=$lib timeout -k 10 120 python scripts/bench_attn.py --tiles $g --recompute 1 --only dq --noscores 2>&1 | grep "^mode" || exit 3; fi
  done
done
