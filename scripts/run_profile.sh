#!/bin/bash
# development aid (run on the GPU box via gpurun): kernel-trace statistics of the headline bench plus the two HBM counter
# passes the roofline's `traffic` is computed from:  run_profile.sh <tag> [bench args, e.g. --config 5 --math fp16]
#   ->  gpurun_out/<tag>/{stats.csv, hbm.txt, bench.json}
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --steps 3 --warmup 1 --headline-only --no-cpu-baseline "$@" > $OUT/kt.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/f -- python3 bench.py --steps 1 --warmup 1 --headline-only --no-cpu-baseline "$@" > $OUT/f.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/w -- python3 bench.py --steps 1 --warmup 1 --headline-only --no-cpu-baseline "$@" > $OUT/w.log 2>&1
cp $(ls $OUT/kt/*/*kernel_stats.csv | head -1) $OUT/stats.csv
{ echo "# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), python3 bench.py --steps 1 --warmup 1 --headline-only --no-cpu-baseline $*";
  echo "# mean per launch, KB; gfx950 HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (MI355X_MICROARCH.md)";
  python3 scripts/pmc_summary.py $OUT/f csn_; python3 scripts/pmc_summary.py $OUT/w csn_; } > $OUT/hbm.txt 2>&1
grep '^{' $OUT/kt.log | tail -1 > $OUT/bench.json   # (the profiler prints after the program's line)
rm -rf $OUT/kt $OUT/f $OUT/w          # the raw traces are large: only the summaries travel back
