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
for n in ${@:-4096 8192}; do
  for who in ref ours; do
    best=""
    for rep in 1 2 3; do
      line=$(OMP_NUM_THREADS=1 $B/$who/init.exe $n | grep "^N=")
      best="$best
$line"
    done
    echo "$who $(echo "$best" | grep N= | sort -t= -k6 -g | head -1)"
  done
done
