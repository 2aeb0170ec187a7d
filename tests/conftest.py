import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # gpu tests fail loudly (not skip) on a GPU box without the HIP path; on CPU-only hosts they are deselected via -m "not gpu"
    pass


@pytest.fixture(scope="session")
def kitti00():
    d = np.load(os.path.join(GOLDEN, "kitti00_dec8.npz"))
    return {k: d[k] for k in d.files}


@pytest.fixture(scope="session")
def kitti07():
    d = np.load(os.path.join(GOLDEN, "kitti07_dec4.npz"))
    return {k: d[k] for k in d.files}


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(GOLDEN, "golden_vgicp.json")) as f:
        return {c["name"]: c for c in json.load(f)["cases"]}


@pytest.fixture(scope="session")
def gpu():
    """The product package on a GPU box; raises (never skips) if the HIP library or the device is missing."""
    import torch

    assert torch.cuda.is_available(), "gpu-marked test running without a GPU"
    import gtsam_points_amd as gpa

    gpa.load()
    return gpa
