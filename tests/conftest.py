import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as orc
    orc.lib()
    return orc


@pytest.fixture(scope="session")
def small_scene():
    from daliti_amd import synth
    return synth.make_small()


@pytest.fixture(scope="session")
def small_tree(oracle, small_scene):
    return oracle.KdTree(small_scene["map"])


def bits(a):
    """View a float array as integers so that equality is bit-exact (and NaN == NaN)."""
    a = np.ascontiguousarray(a)
    return a.view(np.uint32 if a.dtype == np.float32 else np.uint64)
