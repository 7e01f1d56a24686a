#!/bin/bash
# the d = 256 attention launches alone at config-3 geometry under several builds of the library, interleaved, with digests of
# every output:   scripts/dev/ab_attn_libs.sh "nopf hip pf512" [bench_attn.py arguments]
# (variants: scripts/dev/build_variant.sh <name> -D...; "hip" = the production library)
set -e
cd "$(dirname "$0")/../.."
libs=$1; shift
for rep in 1 2; do
  for lib in $libs; do
    echo "## $lib"
    CSN_LIB_PATH=csn_amd/libcsn_$lib.so python scripts/bench_attn.py --tiles --evals 256 --only fwd,dq --digest "$@"
  done
done
