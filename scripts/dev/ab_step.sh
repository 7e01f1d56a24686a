#!/bin/bash
# development aid (GPU box): the step of some configurations with two library builds, interleaved:  ab_step.sh <variant.so> <spec>...
cd "$GRAFT_REPO_ROOT" || exit 1
var=$1; shift
for rep in 1 2; do
  for lib in csn_amd/libcsn_hip.so $var; do
    echo "== $lib"
    CSN_LIB_PATH=$lib bash scripts/dev/quick_bench.sh "$@" || exit 3
  done
done
