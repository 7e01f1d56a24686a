#!/bin/bash
# development aid (GPU box): the flash dK / dV launch at config-5 geometry for several library builds:  ab_dkv.sh <lib.so>...
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do
for lib in "$@"; do
  echo -n "$lib  "
  CSN_LIB_PATH=$lib timeout -k 10 120 python scripts/bench_attn.py --tiles --mode 2 --d 96 --nb 100 --evals 80 --slots 40 --recompute 2 --only dkv --noscores 2>&1 | grep "^mode" || exit 3
done
done
