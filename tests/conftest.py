import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Native pieces are built in-tree once per session (oracle + synth always; the HIP library is
    expected to be prebuilt by __graft_entry__.build() and is only rebuilt when hipcc is present)."""
    from isaac_ros_apriltag_amd import build as b
    from oracle import pyoracle
    b.build_synth()
    pyoracle.lib()
    return True
