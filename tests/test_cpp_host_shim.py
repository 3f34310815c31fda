"""include/brbd_parallel.hpp (C++ mirror of rneaInParallel / abaInParallel / ModelPool) compiles
with g++ against the C-ABI library and keeps the reference's argument checks (host only)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_shim(tmp_path):
    exe = tmp_path / "host_shim_check"
    lib = os.path.join(ROOT, "pinocchio_amd")
    subprocess.check_call(["g++", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "host_shim_check.cpp"), "-o", str(exe),
                           "-L" + lib, "-lbrbd_hip", "-pthread", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ok" in r.stdout
