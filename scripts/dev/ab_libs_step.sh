#!/bin/bash
# whole step under two builds, alternating processes (not interleaved in-process: different libraries)
cd "$GRAFT_REPO_ROOT" || exit 1
for r in 1 2 3; do
  for lib in restage hip; do
    echo "## $lib"
    CSN_LIB_PATH=csn_amd/libcsn_$lib.so python scripts/ab_step.py "$@" --rounds 2 --variants "x:" | grep -v amdgpu
  done
done
