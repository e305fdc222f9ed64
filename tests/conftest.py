import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def pkg():
    import stereo_visual_slam_amd as pkg
    return pkg


@pytest.fixture(scope="session")
def vo(pkg):
    """one device context for the whole GPU session; fails loudly if the HIP library / GPU is missing"""
    ctx = pkg.VO(device=0, max_batch=4)
    yield ctx
    ctx.close()


@pytest.fixture(scope="session")
def synth(pkg):
    from stereo_visual_slam_amd import synth as s
    return s
