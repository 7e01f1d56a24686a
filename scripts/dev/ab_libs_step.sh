#!/bin/bash
# the whole training step under several BUILDS of the library (scripts/dev/build_variant.sh; "hip" = the production library),
# alternating processes:   scripts/dev/ab_libs_step.sh "hip earlyf" --config 3 [--math bf16]
# (compile-time variants cannot be interleaved inside one process like scripts/ab_step.py does for run-time switches: expect a
#  spread of +-0.1 ms between processes and read the alternation, not one pair)
cd "$(dirname "$0")/../.." || exit 1
libs=$1; shift
for r in 1 2 3 4; do
  for lib in $libs; do
    echo "## $lib"
    CSN_LIB_PATH=csn_amd/libcsn_$lib.so python scripts/ab_step.py "$@" --rounds 2 --variants "x:" | grep -v amdgpu
  done
done
