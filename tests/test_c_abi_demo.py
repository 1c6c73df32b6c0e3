"""The C ABI used from a plain C++/HIP host program (examples/c_abi_demo.cpp): it must compile and
link against include/cgps.h + libcgps.so without Python or torch in the picture (CPU check), and
its closed-form checks must pass on the device (GPU check)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "cyclic-gps_amd", "lib")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _build(out):
    if not os.path.exists(os.path.join(LIBDIR, "libcgps.so")):
        subprocess.run([sys.executable, os.path.join(ROOT, "__graft_entry__.py")], check=True, cwd=ROOT)
    cmd = [HIPCC, "-O2", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "c_abi_demo.cpp"), "-L", LIBDIR, "-lcgps",
           "-Wl,-rpath," + LIBDIR, "-o", out]
    subprocess.run(cmd, check=True, cwd=ROOT)
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_demo_compiles_and_links(tmp_path):
    exe = _build(str(tmp_path / "c_abi_demo"))
    assert os.path.getsize(exe) > 0


@pytest.mark.gpu
def test_demo_runs_on_the_device(tmp_path):
    exe = _build(str(tmp_path / "c_abi_demo"))
    for n in ("1", "1000", "300000"):
        r = subprocess.run([exe, n], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "OK (libcgps version" in r.stdout
