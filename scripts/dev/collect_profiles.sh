#!/bin/bash
# development aid (here, after `gpurun -- bash scripts/dev/profile_all.sh <tag>`): copy the summaries of gpurun_out/<tag>_* into
# profiles/ under the round's names and refresh profiles/attn_hbm_traffic.json (what bench.py reports as roofline.traffic)
tag=$1
cd "$(dirname "$0")/../.." || exit 1
for d in gpurun_out/${tag}_c*/; do
  d=${d%/}
  name=${d#gpurun_out/${tag}_}
  cp $d/stats.csv profiles/${tag}_kernel_stats_${name}.csv
  cp $d/hbm.txt profiles/${tag}_pmc_hbm_traffic_${name}.txt
  cp $d/bench.json profiles/${tag}_bench_under_profiler_${name}.json
  cfg=${name%%_*}; math=${name#*_}
  python3 scripts/traffic_json.py profiles/${tag}_pmc_hbm_traffic_${name}.txt ${cfg#c} $math profiles/${tag}_pmc_hbm_traffic_${name}.txt \
      profiles/${tag}_kernel_stats_${name}.csv profiles/${tag}_kernel_stats_${name}.csv > /dev/null
done
ls profiles | grep "^${tag}_" | wc -l
