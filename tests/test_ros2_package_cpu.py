"""The ROS 2 side cannot be built here (no ROS 2 in the image); what can be checked is that the package
description exists with the reference's identity, and that the message definitions carry exactly the fields the
node shell fills (include/apriltag_node_shell.hpp mirrors isaac_ros_apriltag_interfaces; reference
src/apriltag_node.cpp:324-363,500-546, package.xml:34-54)."""
import os
import re
import xml.etree.ElementTree as ET

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R2 = os.path.join(ROOT, "ros2")


def test_package_manifests():
    pkg = ET.parse(os.path.join(R2, "isaac_ros_apriltag", "package.xml")).getroot()
    assert pkg.find("name").text == "isaac_ros_apriltag"
    deps = {d.text for d in pkg.findall("depend")}
    # the non-NVIDIA dependencies of the reference's manifest (package.xml:36-48)
    assert {"isaac_ros_apriltag_interfaces", "message_filters", "rclcpp", "rclcpp_components", "sensor_msgs", "tf2_msgs", "tf2_ros"} <= deps
    assert pkg.find("export/build_type").text == "ament_cmake"
    ipkg = ET.parse(os.path.join(R2, "isaac_ros_apriltag_interfaces", "package.xml")).getroot()
    assert ipkg.find("name").text == "isaac_ros_apriltag_interfaces"
    assert ipkg.find("member_of_group").text == "rosidl_interface_packages"


def _fields(path):
    out = []
    for line in open(path):
        line = line.split("#")[0].strip()
        if line:
            t, n = line.split()[:2]
            out.append((t, n))
    return out


def test_messages_match_the_shell_structs():
    det = _fields(os.path.join(R2, "isaac_ros_apriltag_interfaces", "msg", "AprilTagDetection.msg"))
    arr = _fields(os.path.join(R2, "isaac_ros_apriltag_interfaces", "msg", "AprilTagDetectionArray.msg"))
    assert det == [("string", "family"), ("int32", "id"), ("geometry_msgs/Point", "center"), ("geometry_msgs/Point[4]", "corners"),
                   ("geometry_msgs/PoseWithCovarianceStamped", "pose")]
    assert arr == [("std_msgs/Header", "header"), ("AprilTagDetection[]", "detections")]
    hpp = open(os.path.join(ROOT, "include", "apriltag_node_shell.hpp")).read()
    body = re.search(r"struct AprilTagDetection \{(.*?)\};", hpp, re.S).group(1)
    names = re.findall(r"(\w+)\s*(?:=[^;]*)?;", body)
    assert names == [n for _, n in det]
    body = re.search(r"struct AprilTagDetectionArray \{(.*?)\};", hpp, re.S).group(1)
    assert re.findall(r"(\w+);", body) == [n for _, n in arr]


def test_component_registers_the_reference_plugin_name():
    src = open(os.path.join(R2, "isaac_ros_apriltag", "src", "apriltag_node_component.cpp")).read()
    cm = open(os.path.join(R2, "isaac_ros_apriltag", "CMakeLists.txt")).read()
    assert "nvidia::isaac_ros::apriltag::AprilTagNode" in src and "nvidia::isaac_ros::apriltag::AprilTagNode" in cm
    for prm in ("max_tags", "size", "tile_size", "tag_family", "backends"):     # reference src/apriltag_node.cpp:564-568
        assert 'declare_parameter<' in src and '"%s"' % prm in src
    launch = open(os.path.join(R2, "isaac_ros_apriltag", "launch", "isaac_ros_apriltag.launch.py")).read()
    assert "component_container_mt" in launch and "nvidia::isaac_ros::apriltag::AprilTagNode" in launch
