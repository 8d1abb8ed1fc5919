"""Constructor validation of the node shell -- the three cases of the reference's gtest
(isaac_ros_apriltag/test/apriltag_node_test.cpp:29-89).  Pure host logic, no GPU."""
import pytest

from isaac_ros_apriltag_amd import build as b
from isaac_ros_apriltag_amd import capi

MSG = "Tag family not supported by specified backend"


@pytest.fixture(scope="module")
def node_mod():
    import os
    if not os.path.exists(capi.LIB_PATH):
        b.build_amd()
    b.build_node()
    from isaac_ros_apriltag_amd import node
    node.lib()
    return node


def test_invalid_tag_family(node_mod):
    with pytest.raises(RuntimeError) as e:
        node_mod.AprilTagNode(tag_family="NOTHING")
    assert MSG in str(e.value)


def test_unsupported_tag_family(node_mod):
    # tag36h10 is a family string the reference knows, but this backend has no codebook for it
    with pytest.raises(RuntimeError) as e:
        node_mod.AprilTagNode(tag_family="tag36h10")
    assert MSG in str(e.value) and "'tag_family' parameter must be one of:" in str(e.value)


def test_supported_tag_family_and_defaults(node_mod):
    for fam in ("tag36h11", "tag25h9", "tag16h5"):
        n = node_mod.AprilTagNode(tag_family=fam)
        n.close()
    n = node_mod.AprilTagNode()          # defaults: max_tags 64, size 0.22, tile_size 4, tag36h11
    assert n.max_tags == 64
    n.close()


def test_backend_without_implementation(node_mod):
    # the reference routes 'CPU'/'PVA' to VPI; this build has only the HIP detector and no CPU fallback
    with pytest.raises(RuntimeError) as e:
        node_mod.AprilTagNode(tag_family="tag36h11", backends="CPU")
    assert MSG in str(e.value)
