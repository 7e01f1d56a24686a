#!/bin/bash
# development aid (GPU box): the bench line of every benched configuration / mode -> gpurun_out/<tag>_bench_*.json
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1
timeout -k 10 400 python bench.py 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_config3.json || exit 3
for spec in "config3_bf16:--math bf16" "config2:--config 2" "config2_bf16:--config 2 --math bf16" "config5:--config 5" "config5_fp16:--config 5 --math fp16" "config5_bf16:--config 5 --math bf16" "config3_same_work:--same-work"; do
  name=${spec%%:*}; args=${spec#*:}
  timeout -k 10 300 python bench.py --no-cpu-baseline $args 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_${name}.json || exit 3
done
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/${tag}_bench_*.json")):
    j = json.loads(open(f).read())
    print(f.split("bench_")[1][:-5], round(j["ms_per_step"], 3), j["roofline"]["frac"] if j.get("roofline") else None)
PY
