import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# tools/mutation_check.sh runs parts of the GPU suite against deliberately wrong builds (libapriltag_amd_<tag>.so) to show that the
# suite fails on them; nothing else sets this variable
if os.environ.get("AMDAT_LIB"):
    from isaac_ros_apriltag_amd import capi as _capi
    _capi.LIB_PATH = os.path.join(ROOT, "isaac_ros_apriltag_amd", "libapriltag_amd_%s.so" % os.environ["AMDAT_LIB"])


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
