#!/bin/bash
# FETCH_SIZE (fabric-level read traffic) and duration of the fused kernel under a list of tuning
# settings, one rocprofv3 --pmc pass each.   scripts/fetch_probe.sh T "k=v k=v" "k=v" ...
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/fetch_probe
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
T=$1; shift
n=0
for setting in "$@"; do
    n=$((n+1))
    args=""
    for kv in $setting; do args="$args --tune $kv"; done
    timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/run$n -- python3 bench.py --fused $T --steps $((T*6)) --warmup $T --no-cpu-baseline $args > $OUT/run$n.log 2>&1
    rc=$?
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out - stop"; exit 99; fi
    echo "== [$setting] rc=$rc"
    python scripts/pmc_table.py $OUT/run$n jacobi5xt_tile | tail -2
done
