#!/bin/bash
# development aid: build libcsn variants with extra -D flags:  build_variant.sh <out.so> <flags...>
out=$1; shift
cd /root/repo/csn_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$@" -o $out gemm_f32.hip gemm_bf16x3.hip attn_f32.hip attn_bf16x3.hip attn_dkv.hip outproj_ln.hip retrieval.hip combine.hip compat.hip csn_capi.hip 2>&1 | grep -v "warning: argument unused" || true
