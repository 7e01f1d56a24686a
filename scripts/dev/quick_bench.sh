#!/bin/bash
# development aid (GPU box): the step time of the benched configurations / modes named on the command line, one line each
cd "$GRAFT_REPO_ROOT" || exit 1
for spec in "$@"; do
  args=$(echo "$spec" | tr '_' ' ')
  timeout -k 10 300 python bench.py --no-cpu-baseline --headline-only $args 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read())
ks=[(r['kernel'][:34], round(r['launch_ms'],3)) for r in [j.get('roofline')]+(j.get('roofline_other') if isinstance(j.get('roofline_other'),list) else [j.get('roofline_other')]) if r]
print('$spec', round(j['ms_per_step'],3), j['config'].get('loss'), ks)" || exit 3
done
