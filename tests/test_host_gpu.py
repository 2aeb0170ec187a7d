"""Runs the C++ mirror's end-to-end driver (gtsam_points_amd/host/test_host.cpp) on the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_cpp_host_mirror_end_to_end():
    exe = os.path.join(ROOT, "gtsam_points_amd", "host", "test_host")
    assert os.path.exists(exe), "build it with __graft_entry__.build()"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "HOST_TEST_OK" in out.stdout, out.stdout + out.stderr


def test_cpp_host_mirror_compiles():
    """CPU check: the header-only mirror compiles against the stand-in GTSAM types and links the C-ABI"""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "gtsam_points_amd", "host"), "-s"])
    assert os.path.exists(os.path.join(ROOT, "gtsam_points_amd", "host", "test_host"))
