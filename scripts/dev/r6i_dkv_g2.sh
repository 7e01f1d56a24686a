#!/bin/bash
# round 6: two 16-key groups per wave in the key-stationary dK / dV kernel (libcsn_hip.so) against one (libcsn_g1.so, -DCSN_DKV_G2=0)
# and the round-5 kernel (libcsn_old.so)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$1; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_flash.py tests/test_gpu_act16.py tests/test_gpu_lowprec.py tests/test_gpu_configs.py -m gpu -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
bash scripts/dev/ab_libs_step.sh "old g1 hip" --config 5 --math fp16 > $O/step_c5_fp16.txt 2>&1 || exit 1
grep -E "##|median" $O/step_c5_fp16.txt | paste - - | sed 's/gradients vs x: bit-equal//'
bash scripts/dev/ab_libs_step.sh "old g1 hip" --config 5 --math bf16 > $O/step_c5_bf16.txt 2>&1 || exit 1
grep -E "##|median" $O/step_c5_bf16.txt | paste - - | sed 's/gradients vs x: bit-equal//'
