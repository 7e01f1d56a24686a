#!/bin/bash
# Build csn_amd/libcsn_<name>.so from the same sources with extra compiler flags (timing experiments; load it with
# CSN_LIB_PATH=csn_amd/libcsn_<name>.so).   scripts/dev/build_variant.sh rcx -DCSN_RC_ALIAS=1
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
obj=csn_amd/_obj_$name
mkdir -p $obj
pids=()
for f in csn_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $f -o $obj/$b.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o csn_amd/libcsn_$name.so $obj/*.o
echo built csn_amd/libcsn_$name.so
