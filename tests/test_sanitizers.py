"""AddressSanitizer + UndefinedBehaviorSanitizer over the host-side C++ (CPU build only: GPU sanitizers are not
available on this pool): oracle modes A/B, host mirror, product BVH builder."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_host_cpp_under_asan_ubsan(tmp_path):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    exe = str(tmp_path / "sanitize_main")
    cmd = [gxx, "-std=c++17", "-O1", "-g", "-fopenmp", "-ffp-contract=off", "-mavx2", "-mfma",
           "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-o", exe,
           os.path.join(HERE, "sanitize_main.cpp")]
    b = subprocess.run(cmd, capture_output=True, text=True)
    if b.returncode != 0 and ("asan" in b.stderr.lower() or "ubsan" in b.stderr.lower()) and "cannot find" in b.stderr:
        pytest.skip("sanitizer runtimes not installed")
    assert b.returncode == 0, b.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
               OMP_NUM_THREADS="2")
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "sanitizer run ok" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
