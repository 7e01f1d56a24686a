#!/bin/bash
# development aid (GPU box): kernel statistics + HBM counter passes of every benched configuration / mode -> gpurun_out/<tag>_*
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1
for spec in "c3_bf16x3:" "c3_bf16:--math bf16" "c2_bf16:--config 2 --math bf16" "c5_fp16:--config 5 --math fp16" "c5_bf16x3:--config 5" "c2_bf16x3:--config 2"; do
  name=${spec%%:*}; args=${spec#*:}
  bash scripts/run_profile.sh ${tag}_${name} $args || exit 3
done
