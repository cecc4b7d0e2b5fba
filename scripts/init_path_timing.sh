#!/bin/bash
# the init path (grid_init + four r2d_field constructors) of the REAL reference build (oracle/_ref, serial
# sources compiled in place by oracle/Makefile) timed beside this repository's Fortran layer, in THIS container (the
# reference does not travel to the GPU box).  One rank, one thread (the reference declares OpenMP builds unsupported,
# field_mod.f90:302-303); best of three runs per size.  Output: a small table on stdout (-> BASELINE.md).
#
#   scripts/init_path_timing.sh [sizes ...]        default: 4096 8192
set -eu
cd "$(dirname "$0")/.."
ROOT=$PWD
[ -f oracle/_ref/lib_fd.a ] || make -C oracle ref > /dev/null
[ -f dl_esm_inf_amd/fortran/build/lib_fd_hip.a ] || make -C dl_esm_inf_amd/fortran lib > /dev/null
B=scratch/init_timing
mkdir -p $B/ref $B/ours
amdflang -O2 -I oracle/_ref/obj -J $B/ref scripts/init_path_timing.f90 oracle/_ref/lib_fd.a -o $B/ref/init.exe
amdflang -O2 -I dl_esm_inf_amd/fortran/build -J $B/ours scripts/init_path_timing.f90 dl_esm_inf_amd/fortran/build/lib_fd_hip.a \
    -L dl_esm_inf_amd/lib -ldlesm_hip -L/opt/rocm/lib -lamdhip64 -lrccl -Wl,-rpath,$ROOT/dl_esm_inf_amd/lib -Wl,-rpath,/opt/rocm/lib \
    -o $B/ours/init.exe
echo "host: $(nproc) cores, $(grep -m1 'model name' /proc/cpuinfo | cut -d: -f2 | sed 's/^ //'); 1 rank, 1 thread; DL_ESM_ALIGNMENT=${DL_ESM_ALIGNMENT:-unset}"
REPS=${REPS:-5}
printf "%-5s %6s  %-22s %-22s %s\n" build N "grid_init s (min-max)" "four fields s (min-max)" "checksum of a field of ones"
for n in ${@:-4096 8192}; do
  for who in ref ours; do          # alternating runs, $REPS each: page-fault cost in this VM varies a lot from run to run
    : > $B/$who.$n.txt
  done
  for rep in $(seq $REPS); do
    for who in ref ours; do
      OMP_NUM_THREADS=1 $B/$who/init.exe $n | grep "^N=" >> $B/$who.$n.txt
    done
  done
  for who in ref ours; do
    python3 - $who $n $B/$who.$n.txt <<'PY'
import re, sys
who, n, f = sys.argv[1:]
g, ff, cs = [], [], set()
for line in open(f):
    m = re.search(r"grid_init_s=\s*([0-9.]+) four_fields_s=\s*([0-9.]+) checksum=\s*(\S+)", line)
    g.append(float(m.group(1))); ff.append(float(m.group(2))); cs.add(m.group(3))
print(f"{who:5s} {n:>6s}  {min(g):8.3f} - {max(g):8.3f}    {min(ff):8.3f} - {max(ff):8.3f}    {' '.join(sorted(cs))}")
PY
  done
done
