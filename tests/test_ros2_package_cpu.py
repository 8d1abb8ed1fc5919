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


# ---- the rclcpp component parses and type-checks against the shell header (stand-in ROS 2 headers) ----------------------
_CPP_TYPES = {"string": "std::string", "int32": "int32_t", "uint32": "uint32_t", "float64": "double", "bool": "bool"}


def _msg_to_cpp(pkg, name, path):
    """One .msg file -> a C++ struct with the field names and container shapes rosidl would generate."""
    incs, fields = {"<array>", "<cstdint>", "<string>", "<vector>"}, []
    for t, n in _fields(path):
        m = re.match(r"^([\w/]+)(\[(\d*)\])?$", t)
        base, arr, cnt = m.group(1), m.group(2), m.group(3)
        if "/" in base:
            p2, b2 = base.split("/")
            snake = re.sub(r"(?<!^)(?=[A-Z])", "_", b2).lower()
            incs.add('"%s/msg/%s.hpp"' % (p2, snake))
            ct = "%s::msg::%s" % (p2, b2)
        elif base in _CPP_TYPES:
            ct = _CPP_TYPES[base]
        else:   # a message of the same package
            snake = re.sub(r"(?<!^)(?=[A-Z])", "_", base).lower()
            incs.add('"%s/msg/%s.hpp"' % (pkg, snake))
            ct = "%s::msg::%s" % (pkg, base)
        if arr:
            ct = "std::array<%s, %s>" % (ct, cnt) if cnt else "std::vector<%s>" % ct
        fields.append("  %s %s{};" % (ct, n))
    return "#pragma once\n%s\nnamespace %s { namespace msg { struct %s {\n%s\n}; } }\n" % (
        "\n".join("#include %s" % i for i in sorted(incs)), pkg, name, "\n".join(fields))


def test_component_parses_against_shell_header(tmp_path):
    """g++ -fsyntax-only on ros2/isaac_ros_apriltag/src/apriltag_node_component.cpp (mirror of the reference's
    src/apriltag_node.cpp:562-633) against tests/aux_c/ros2_stubs (stand-ins for rclcpp, message_filters, tf2_ros and the
    common message types; the isaac_ros_apriltag_interfaces headers are generated here from the package's .msg files) and
    the real include/apriltag_node_shell.hpp: every field the adapter copies exists on both sides with a convertible type,
    the shell's callbacks and options have the signatures the adapter uses, and the class is a constructible component."""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        import pytest
        pytest.skip("no g++")
    gen = tmp_path / "isaac_ros_apriltag_interfaces" / "msg"
    gen.mkdir(parents=True)
    mdir = os.path.join(R2, "isaac_ros_apriltag_interfaces", "msg")
    (gen / "april_tag_detection.hpp").write_text(_msg_to_cpp("isaac_ros_apriltag_interfaces", "AprilTagDetection", os.path.join(mdir, "AprilTagDetection.msg")))
    (gen / "april_tag_detection_array.hpp").write_text(_msg_to_cpp("isaac_ros_apriltag_interfaces", "AprilTagDetectionArray", os.path.join(mdir, "AprilTagDetectionArray.msg")))
    src = os.path.join(R2, "isaac_ros_apriltag", "src", "apriltag_node_component.cpp")
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "tests", "aux_c", "ros2_stubs"),
           "-I", str(tmp_path), "-I", os.path.join(ROOT, "include"), src]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    # and the check has teeth: a field the .msg files do not define must not compile
    broken = tmp_path / "broken.cpp"
    broken.write_text(open(src).read().replace("m.family = d.family;", "m.family_name = d.family;"))
    r2 = subprocess.run(cmd[:-1] + [str(broken)], capture_output=True, text=True)
    assert r2.returncode != 0 and "family_name" in r2.stderr
