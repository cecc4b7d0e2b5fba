#!/bin/bash
# Fortran-hosted timings of the round-3 code (north_star: host code in Fortran over ISO_C_BINDING), next to the
# Python/ctypes-hosted figures of the same box:   scripts/fortran_host_bench.sh > profiles/r03_fortran_host.txt
set -u
cd "$(dirname "$0")/.."
B=dl_esm_inf_amd/fortran/build
export DL_ESM_ALIGNMENT=64
unset RANK WORLD_SIZE LOCAL_RANK
echo "# Fortran host (amdflang programs on lib_fd_hip.a -> libdlesm_hip.so), DL_ESM_ALIGNMENT=64, one MI355X"
echo "# jacobi_app.exe N NSTEPS FUSE PLAN : 10 untimed warm-up steps, then NSTEPS timed steps (system_clock around the loop + device_sync)"
for cfg in "4096 2000" "8192 800" "16384 400"; do
    set -- $cfg
    for plan in 0 1; do
        for rep in 1 2; do
            echo -n "jacobi_app $1^2 steps=$2 plan=$plan run=$rep : "
            timeout -k 10 200 $B/jacobi_app.exe $1 $2 1 $plan 2>&1 | grep -E "Mcells" | tr -s ' '
        done
    done
done
echo "# shallow_app.exe N NSTEPS MODE : 0 = fused one-launch periodic step, 1 = the seven GOcean kernels one by one (+ periodic copies), 2 = 1 + time_smooth, 3 = one launch per step incl. time_smooth, 4 = one launch per TWO steps incl. time_smooth"
for mode in 0 1 2 3 4; do
    for rep in 1 2; do
        echo -n "shallow_app 8192^2 steps=200 mode=$mode run=$rep : "
        timeout -k 10 200 $B/shallow_app.exe 8192 200 $mode 2>&1 | grep -E "Mcells" | tr -s ' '
    done
done
echo "# the same kernels hosted by Python/ctypes (bench.py, same box, same process launch):"
timeout -k 10 400 python bench.py --no-cpu-baseline --no-temporal-blocking --no-weak-tile > /tmp/bench_fh.json 2>/dev/null
python - <<'PY'
import json
d = json.load(open("/tmp/bench_fh.json"))
print(f"bench.py 16384^2 A=64 planned : {d['value']:.1f} Mcells/s  ({d['roofline']['frac']:.4f} of peak)")
for c in d.get("configs", []):
    print(f"bench.py {c['tile']}^2 A={c['DL_ESM_ALIGNMENT']} planned : {c['value']:.1f} Mcells/s")
sw = d["shallow_water"]
print(f"bench.py shallow fused NE 8192^2 planned : {sw['value']:.1f} Mcells/s; SW periodic one launch : {sw['sw_offset_periodic']['value']:.1f}; "
      f"seven kernels (NE, no periodic copies) : {sw['unfused']['value']:.1f}")
PY
