#!/bin/bash
# One gpurun call = one batch of GPU work (getting a box costs minutes, so batch).
# Every step runs under its own timeout; a step that TIMES OUT (rc 124/137) stops the batch
# (no further GPU step after a hang), an ordinary failure is logged and the batch goes on.
#
#   scripts/gpu_batch.sh tests sweep prof pmc bench      (any subset, in that order)
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp

step() { # name timeout cmd...
    local name=$1 t=$2; shift 2
    echo "=== [$name] $(date +%T) : $*" >&2
    timeout -k 10 "$t" "$@"
    local rc=$?
    echo "=== [$name] rc=$rc" >&2
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then
        echo "=== [$name] TIMED OUT - stopping the batch" >&2; exit 99
    fi
    return $rc
}

for what in "$@"; do
case $what in
tests)
    step tests 1000 python -m pytest tests -m gpu -q -x --timeout=600 > $OUT/gpu_tests.log 2>&1
    tail -15 $OUT/gpu_tests.log ;;
prof)
    rm -rf $OUT/prof_stats
    step prof 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python3 bench.py --no-cpu-baseline > $OUT/prof_stats.log 2>&1
    tail -2 $OUT/prof_stats.log
    python scripts/parse_rocprof.py stats $OUT/prof_stats $OUT/prof_stats_summary.md > /dev/null 2>&1 || echo "parse failed" ;;
pmc)
    rm -rf $OUT/pmc_fetch $OUT/pmc_write
    [ -f profiles/traffic.json ] && cp profiles/traffic.json $OUT/traffic.json   # records are merged by key
    step pmcF 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-temporal-blocking --no-shallow --no-configs --no-weak-tile --no-peer > $OUT/pmc_fetch.log 2>&1
    step pmcW 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-temporal-blocking --no-shallow --no-configs --no-weak-tile --no-peer > $OUT/pmc_write.log 2>&1
    python scripts/parse_rocprof.py pmc $OUT/pmc_fetch $OUT/pmc_write "16384x16384/A64" $OUT/traffic.json jacobi5_ 20 2>&1 | tail -12 ;;   # the 20 timed launches (planned shape)
pmcx)    # HBM traffic of the fused kernel, FUSED steps per launch (default 8)
    F=${FUSED:-8}
    rm -rf $OUT/pmcx_fetch $OUT/pmcx_write
    [ -f profiles/traffic.json ] && cp profiles/traffic.json $OUT/traffic.json   # records are merged by key
    step pmcxF 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmcx_fetch -- python3 bench.py --fused $F --steps $((F*5)) --warmup $F --no-cpu-baseline > $OUT/pmcx_fetch.log 2>&1
    step pmcxW 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmcx_write -- python3 bench.py --fused $F --steps $((F*5)) --warmup $F --no-cpu-baseline > $OUT/pmcx_write.log 2>&1
    python scripts/parse_rocprof.py pmc $OUT/pmcx_fetch $OUT/pmcx_write "16384x16384/A64/fused$F" $OUT/traffic.json jacobi5xt_ 2>&1 | tail -12 ;;
benchx)
    F=${FUSED:-8}
    step benchx 400 python bench.py --fused $F --steps $((F*25)) --no-cpu-baseline > $OUT/bench_fused$F.json 2> $OUT/bench_fused$F.err
    cat $OUT/bench_fused$F.json ;;
bench)
    step bench 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err
    cat $OUT/bench.json ;;
smoke)
    step smoke 200 python __graft_entry__.py smoke ;;
probe)
    rm -rf $OUT/probe_fetch $OUT/probe_write
    step probeF 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/probe_fetch -- python3 scripts/pmc_probe.py > $OUT/probe_fetch.log 2>&1
    step probeW 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/probe_write -- python3 scripts/pmc_probe.py > $OUT/probe_write.log 2>&1
    python scripts/parse_probe.py $OUT/probe_fetch $OUT/probe_write > $OUT/probe_table.txt 2>&1
    cat $OUT/probe_table.txt ;;
dm)
    step dm8192 300 python scripts/dm_overhead.py --tile 8192 --out $OUT/dm_overhead_8192.json > $OUT/dm_overhead_8192.log 2>&1
    tail -22 $OUT/dm_overhead_8192.log
    step dm16384 300 python scripts/dm_overhead.py --tile 16384 --out $OUT/dm_overhead_16384.json > $OUT/dm_overhead_16384.log 2>&1
    tail -22 $OUT/dm_overhead_16384.log ;;
dmfused)
    for t in 8192 16384; do
        F=${FUSED:-8}
        step dmf$t 300 python scripts/dm_overhead.py --fused $F --tile $t --steps 40 --out $OUT/dm_fused${F}_$t.json > $OUT/dm_fused${F}_$t.log 2>&1
        grep -v amdgpu.ids $OUT/dm_fused${F}_$t.log | tail -20
    done ;;
shallow)
    step shallow 400 python scripts/shallow_bench.py --out $OUT/shallow_bench.json > $OUT/shallow_bench.log 2>&1
    grep -v amdgpu.ids $OUT/shallow_bench.log | tail -8 ;;
shallowpmc)
    rm -rf $OUT/swpmc_fetch $OUT/swpmc_write $OUT/sw_prof
    step swF 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/swpmc_fetch -- python3 scripts/shallow_bench.py --steps 3 --no-cpu --out $OUT/sw_tmp.json > $OUT/swpmc_fetch.log 2>&1
    step swW 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/swpmc_write -- python3 scripts/shallow_bench.py --steps 3 --no-cpu --out $OUT/sw_tmp.json > $OUT/swpmc_write.log 2>&1
    python scripts/parse_rocprof.py pmc $OUT/swpmc_fetch $OUT/swpmc_write "shallow 8192x8192/A64" $OUT/traffic_shallow.json shallow_tile 2>&1 | tail -12
    step swP 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sw_prof -- python3 scripts/shallow_bench.py --steps 10 --no-cpu --out $OUT/sw_tmp.json > $OUT/sw_prof.log 2>&1
    python scripts/parse_rocprof.py stats $OUT/sw_prof $OUT/sw_prof_summary.md | cut -c1-170 | tail -8 ;;
swkpmc)   # fabric traffic of the seven GOcean kernels launched one by one (and time_smooth), 8192^2
    rm -rf $OUT/swk_fetch $OUT/swk_write
    step swkF 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/swk_fetch -- python3 scripts/shallow_r3_probe.py --what kernels --steps 4 --passes 1 --out $OUT/swk_tmp.json > $OUT/swk_fetch.log 2>&1
    step swkW 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/swk_write -- python3 scripts/shallow_r3_probe.py --what kernels --steps 4 --passes 1 --out $OUT/swk_tmp.json > $OUT/swk_write.log 2>&1
    rm -f $OUT/traffic_swk.json
    for k in CuNE CvNE ZNE HNE UnewNE VnewNE PnewNE TimeSmooth; do
        python scripts/parse_rocprof.py pmc $OUT/swk_fetch $OUT/swk_write "swk $k 8192x8192/A64" $OUT/traffic_swk.json "::$k," 2>&1 | grep -E "hbm_bytes_per_launch|no counter"
    done ;;
swcounters)   # occupancy / VALU / L2 counters of shallow_tile, one counter per pass (bench.py's shallow leg: 8192^2)
    rm -rf $OUT/swcounters
    for cnt in ${PMC_LIST:-VALUBusy MemUnitStalled MeanOccupancyPerActiveCU TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU FETCH_SIZE WRITE_SIZE}; do
        step swc_$cnt 200 rocprofv3 --pmc $cnt --output-format csv -d $OUT/swcounters/$cnt -- python3 scripts/shallow_bench.py --steps 3 --no-cpu --only-default --out $OUT/sw_tmp.json > $OUT/swcounters_$cnt.log 2>&1
    done
    python scripts/pmc_table.py $OUT/swcounters shallow_tile > $OUT/swcounters_table.txt 2>&1
    cat $OUT/swcounters_table.txt ;;
pmcconfigs)   # fabric traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the other BASELINE Jacobi configurations
    [ -f $OUT/traffic.json ] || { [ -f profiles/traffic.json ] && cp profiles/traffic.json $OUT/traffic.json; }
    for cfg in "8192 64" "4096 64" "16384 1" "4096 1"; do
        set -- $cfg
        rm -rf $OUT/pmcc_fetch $OUT/pmcc_write
        step pmccF$1 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmcc_fetch -- python3 bench.py --tile $1 --alignment $2 --steps 20 --warmup 4 --no-cpu-baseline --no-temporal-blocking --no-shallow --no-configs --no-weak-tile --no-peer > $OUT/pmcc_fetch.log 2>&1
        step pmccW$1 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmcc_write -- python3 bench.py --tile $1 --alignment $2 --steps 20 --warmup 4 --no-cpu-baseline --no-temporal-blocking --no-shallow --no-configs --no-weak-tile --no-peer > $OUT/pmcc_write.log 2>&1
        python scripts/parse_rocprof.py pmc $OUT/pmcc_fetch $OUT/pmcc_write "$1x$1/A$2" $OUT/traffic.json jacobi5_ 20 2>&1 | tail -4
    done ;;
dmprof)
    rm -rf $OUT/dm_prof
    step dmprof 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dm_prof -- python3 scripts/dm_overhead.py --tile ${DM_TILE:-16384} --steps 20 --out $OUT/dm_overhead_prof.json > $OUT/dm_prof.log 2>&1
    python scripts/parse_rocprof.py stats $OUT/dm_prof $OUT/dm_prof_summary.md | cut -c1-170 | tail -14 ;;
configs)
    : > $OUT/configs.jsonl
    for cfg in "4096 64" "8192 64" "16384 64" "16384 1" "4096 1"; do
        set -- $cfg
        step "cfg$1a$2" 300 python bench.py --tile $1 --alignment $2 --steps 100 --warmup 10 --cpu-seconds 4 >> $OUT/configs.jsonl 2>> $OUT/configs.err
    done
    python - <<'PYEOF'
import json
for l in open("gpurun_out/configs.jsonl"):
    if not l.startswith("{"): continue
    d = json.loads(l)
    c = d["cpu_baseline"] or {}
    print(d["config"]["tile"], d["config"]["DL_ESM_ALIGNMENT"], d["config"]["ld"], d["value"], d["hbm_gbs_per_gpu"],
          d["roofline"]["frac"], c.get("value"), c.get("single_core_value"))
PYEOF
    ;;
tune)
    # TUNE_ARGS: the --grid ... arguments of scripts/sweep_tune.py
    step tune 500 python scripts/sweep_tune.py ${TUNE_ARGS} --out $OUT/sweep_tune.json > $OUT/sweep_tune.log 2>&1
    head -${SWEEP_HEAD:-40} $OUT/sweep_tune.log ;;
membench)
    step membench 300 ./build/membench > $OUT/membench.log 2>&1
    cat $OUT/membench.log ;;
pmcmulti)   # one counter per pass, default bench (single-step kernel + the fused-4 secondary leg)
    rm -rf $OUT/pmcmulti
    for cnt in ${PMC_LIST:-VALUBusy MemUnitStalled TCC_HIT_sum TCC_MISS_sum MeanOccupancyPerActiveCU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU TCP_TCC_READ_REQ_sum}; do
        step pmc_$cnt 200 rocprofv3 --pmc $cnt --output-format csv -d $OUT/pmcmulti/$cnt -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > $OUT/pmcmulti_$cnt.log 2>&1
    done
    python scripts/pmc_table.py $OUT/pmcmulti jacobi5_tile jacobi5xt_tile > $OUT/pmcmulti_table.txt 2>&1
    cat $OUT/pmcmulti_table.txt ;;
pmccmp)     # counters of the linear-sweep tile kernel against the column-marching pipeline kernel, one step each
    rm -rf $OUT/pmccmp
    for cnt in ${PMC_LIST:-TCP_PENDING_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_STALL_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum MemUnitStalled TA_BUSY_avr TCC_BUSY_avr}; do
        step a_$cnt 200 rocprofv3 --pmc $cnt --output-format csv -d $OUT/pmccmp/tile/$cnt -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-temporal-blocking > $OUT/pmccmp_a.log 2>&1
        step b_$cnt 200 rocprofv3 --pmc $cnt --output-format csv -d $OUT/pmccmp/march/$cnt -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-temporal-blocking --tune j5_kernel=3 --tune j5xt_march=1 > $OUT/pmccmp_b.log 2>&1
    done
    python scripts/pmc_table.py $OUT/pmccmp jacobi5_tile jacobi5xt_march > $OUT/pmccmp_table.txt 2>&1
    cat $OUT/pmccmp_table.txt ;;
shapes)     # exhaustive (waves per group, tiles per row) search per size + what the rule picks
    : > $OUT/shape_search.txt
    for t in ${SIZES:-1024 2048 3072 4096 5000 7000 8192 10000 12288 14000 16384}; do
        step shape$t 200 python scripts/shape_search.py $t 2>&1 | grep -v amdgpu.ids | grep -E "^N |best:" | head -5 >> $OUT/shape_search.txt
    done
    cat $OUT/shape_search.txt ;;
counters)
    step counters 120 rocprofv3 -L > $OUT/counters.txt 2>&1
    grep -c . $OUT/counters.txt ;;
*)  echo "unknown step $what" ;;
esac
done
ls -la $OUT | head -30
