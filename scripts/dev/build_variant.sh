#!/bin/bash
# Build csn_amd/libcsn_<name>.so from the same sources with extra compiler flags (timing experiments; load it with
# CSN_LIB_PATH=csn_amd/libcsn_<name>.so).   scripts/dev/build_variant.sh rcx -DCSN_RC_ALIAS=1
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
obj=csn_amd/_obj_$name
mkdir -p $obj
pids=()
for f in $(python3 -c "from csn_amd import _lib; print(' '.join('csn_amd/csrc/' + f for f in _lib.SOURCES))"); do
  b=$(basename $f .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden "$@" -c $f -o $obj/$b.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -shared -o csn_amd/libcsn_$name.so $obj/*.o
echo built csn_amd/libcsn_$name.so
