"""The C ABI driven from C++ (the reference's host language), no Python in the loop."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "odometry_demo")
    lib = os.path.join(ROOT, "daliti_amd", "_lib")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "odometry_demo.cpp"), "-L", lib, "-ldaliti_s2m",
                           "-Wl,-rpath," + lib, "-o", exe])
    return exe


def test_cpp_demo_links_against_the_abi_and_fails_loudly_without_gpu(tmp_path):
    import torch
    exe = _build(tmp_path)          # plain g++: the header is C, no HIP or torch types leak through it
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([exe, "1"], capture_output=True, text=True)
    assert r.returncode == 2 and "no gfx950 HIP device" in r.stderr


@pytest.mark.gpu
def test_cpp_demo_tracks_the_motion(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe, "12"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "worst position error" in r.stdout
