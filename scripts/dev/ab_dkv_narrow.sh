#!/bin/bash
# development aid (GPU box): the flash dK / dV launch at d = 64 and d = 32 (one plane, 16-bit and fp32 maps come out the same way) for several builds
cd "$GRAFT_REPO_ROOT" || exit 1
for g in "--d 64 --evals 256" "--d 32 --evals 256"; do
for rep in 1 2; do
for lib in "$@"; do
  echo -n "$lib $g  "
  CSN_LIB_PATH=$lib timeout -k 10 120 python scripts/bench_attn.py --tiles --mode 2 $g --recompute 2 --only dkv --noscores 2>&1 | grep "dkv" || exit 3
done
done
done
