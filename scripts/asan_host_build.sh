#!/bin/bash
# Host-side AddressSanitizer + UBSan run of the C-ABI library (the GPU pool has no device ASan: sanitizers run on the CPU build
# only).  Every source is compiled with -fsanitize=address,undefined for the HOST half only (-fno-gpu-sanitize), linked into
# scratch/asan/libdlesm_hip_asan.so, and the CPU test suite (index maps, decomposition, bounds, comm tables, the rendezvous board,
# argument checking of every entry, the Fortran layer's host paths) runs against it.   scripts/asan_host_build.sh
set -eu
cd "$(dirname "$0")/.."
OUT=scratch/asan
mkdir -p $OUT
FL="-O1 -g -std=c++17 -fPIC -ffp-contract=off -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer --offload-arch=gfx950 -Iinclude"
for f in dl_esm_inf_amd/csrc/*.cpp; do /opt/rocm/bin/hipcc $FL -x hip -c $f -o $OUT/$(basename $f).o; done
ls dl_esm_inf_amd/csrc/*.hip | xargs -P 6 -I{} sh -c "/opt/rocm/bin/hipcc $FL -c {} -o $OUT/\$(basename {}).o"
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -fsanitize=address,undefined -shared-libsan -o $OUT/libdlesm_hip_asan.so $OUT/*.o -L/opt/rocm/lib -lrccl
ASAN_LIB=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
rm -f /tmp/dlesm_asan_log* /tmp/dlesm_ubsan_log*
# (the reference's own test_device_io, run as a child by tests/test_reference_programs_dropin.py, writes through a pointer its main
#  program has already deallocated -- a finding in the REFERENCE's test under ASan, not in this library: deselected here)
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:detect_odr_violation=0:log_path=/tmp/dlesm_asan_log \
UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0:log_path=/tmp/dlesm_ubsan_log \
DLESM_HIP_LIB=$PWD/$OUT/libdlesm_hip_asan.so LD_PRELOAD=$ASAN_LIB \
python -m pytest tests -q -m "not gpu" -p no:cacheprovider \
    --deselect tests/test_cabi_host.py::test_product_and_lab_build_export_the_same_entries \
    --deselect tests/test_reference_programs_dropin.py::test_reference_device_io_test
echo "sanitizer reports: $(ls /tmp/dlesm_asan_log* /tmp/dlesm_ubsan_log* 2>/dev/null | wc -l)"
