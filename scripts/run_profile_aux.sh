#!/bin/bash
# development aid (GPU box): kernel statistics + HBM counter passes of the two paths beside the training step —
# the retrieval measure behind the kNN graph (K7, MID-FC/csa_models.py:244-267) and the MinkowskiNet variant of the layer (f2,
# MinkowskiNet/models/attention.py:31-56):  run_profile_aux.sh <tag>  ->  gpurun_out/<tag>_{retrieval,minkowski}.{txt,csv,hbm.txt}
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
tag=$1
for what in retrieval minkowski; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/${tag}_$what; mkdir -p $OUT
  timeout -k 10 200 python3 scripts/bench_$what.py > $OUT/bench.txt 2>&1 || exit 3
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 scripts/bench_$what.py > $OUT/kt.log 2>&1 &&
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/f -- python3 scripts/bench_$what.py > $OUT/f.log 2>&1 &&
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/w -- python3 scripts/bench_$what.py > $OUT/w.log 2>&1 || exit 4
  cp $(ls $OUT/kt/*/*kernel_stats.csv | head -1) $OUT/stats.csv
  { echo "# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), python3 scripts/bench_$what.py";
    echo "# mean per launch, KB; gfx950 HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (MI355X_MICROARCH.md)";
    python3 scripts/pmc_summary.py $OUT/f csn_; python3 scripts/pmc_summary.py $OUT/w csn_; } > $OUT/hbm.txt 2>&1
  rm -rf $OUT/kt $OUT/f $OUT/w
done
