"""The C++ drop-in adapter (include/morb/ORBextractor.h) compiled with g++ and linked to libmorb_hip.so."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_adapter_builds_and_runs(tmp_path):
    exe = str(tmp_path / "adapter_smoke")
    libdir = os.path.join(ROOT, "morb_slam_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), "-o", exe,
                           os.path.join(ROOT, "tests", "native", "adapter_smoke.cc"), "-L" + libdir, "-lmorb_hip",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "adapter smoke" in out.stdout
