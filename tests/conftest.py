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
def oracle_mod():
    import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def ctx():
    import image_stitching_amd as isa
    c = isa.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def small_pair():
    """Two overlapping 480x270 synthetic frames (yaw 0 / 12 deg) with cameras."""
    import synth
    cams = [synth.make_camera(480, 270, 60.0, 0.0), synth.make_camera(480, 270, 60.0, 12.0, 0.7, -0.4)]
    return cams, [synth.render_frame(c) for c in cams]
