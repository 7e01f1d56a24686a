#!/bin/bash
# Build csn_amd/libcsn_<name>.so = the production objects (csn_amd/_obj, built by csn_amd.build()) with ONE source recompiled
# under extra compiler flags (timing experiments; load it with CSN_LIB_PATH=csn_amd/libcsn_<name>.so):
#   scripts/dev/build_variant_one.sh abl2 attn_dkv.hip -DCSN_DKV_ABL=2
set -e
cd "$(dirname "$0")/../.."
name=$1; src=$2; shift; shift
python3 -c "import csn_amd; csn_amd.build()"
obj=csn_amd/_obj_$name
mkdir -p $obj
b=$(basename $src .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden "$@" -c csn_amd/csrc/$src -o $obj/$b.o
objs=$(python3 -c "from csn_amd import _lib; print(' '.join(('$obj/' if f == '$src' else 'csn_amd/_obj/') + f.replace('.hip', '.o') for f in _lib.SOURCES))")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -shared -o csn_amd/libcsn_$name.so $objs
echo built csn_amd/libcsn_$name.so
