"""AddressSanitizer + UBSan over the host-only parts of the library (CPU build of
dl_esm_inf_amd/csrc/dlesm_maps.cpp and dlesm_rendezvous.cpp, harness tests/sanitize_maps.cpp)."""
import os
import subprocess

from conftest import ROOT


def test_index_maps_under_asan_ubsan(tmp_path):
    exe = tmp_path / "sanitize_maps"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "dl_esm_inf_amd", "csrc"),
           os.path.join(ROOT, "dl_esm_inf_amd", "csrc", "dlesm_maps.cpp"),
           os.path.join(ROOT, "dl_esm_inf_amd", "csrc", "dlesm_rendezvous.cpp"),
           os.path.join(ROOT, "tests", "sanitize_maps.cpp"), "-o", str(exe)]
    subprocess.check_call(cmd)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    p = subprocess.run([str(exe)], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "checks passed" in p.stdout
