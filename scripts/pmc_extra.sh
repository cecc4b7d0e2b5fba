set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out
rm -rf $OUT/px_fetch $OUT/px_write $OUT/py_fetch $OUT/py_write
cp profiles/traffic.json $OUT/traffic_extra.json
timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/px_fetch -- python3 scripts/dm_overhead.py --tile 8192 --steps 20 --out $OUT/dm_tmp.json > $OUT/px_fetch.log 2>&1
timeout -k 10 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/px_write -- python3 scripts/dm_overhead.py --tile 8192 --steps 20 --out $OUT/dm_tmp.json > $OUT/px_write.log 2>&1
python scripts/parse_rocprof.py pmc $OUT/px_fetch $OUT/px_write "8192x8192/A64/framed (frame workgroups + interior, RCCL loop-back)" $OUT/traffic_extra.json jacobi5_tile_framed 2>&1 | tail -5
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/py_fetch -- python3 scripts/shallow_bench.py --steps 5 --no-cpu --out $OUT/sw_tmp.json > $OUT/py_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/py_write -- python3 scripts/shallow_bench.py --steps 5 --no-cpu --out $OUT/sw_tmp.json > $OUT/py_write.log 2>&1
python scripts/parse_rocprof.py pmc $OUT/py_fetch $OUT/py_write "shallow SW-offset 8192x8192/A64" $OUT/traffic_extra.json shallow_tile_sw 2>&1 | tail -5
