"""Loads the AprilTag component into a multi-threaded container.

Plugin identity, node name, container type and the three fixed parameters (size, max_tags, tile_size) are those of the
reference's stand-alone launch file (launch/isaac_ros_apriltag.launch.py:25-40), which subscribes to `image` /
`camera_info` unremapped.  The launch arguments `tag_family` / `backends` and the optional remapping onto the rectified
topics (`image` -> `image_rect`, `camera_info` -> `camera_info_rect`) come from the reference's CORE fragment
(launch/isaac_ros_apriltag_core.launch.py:33-69), the one its composed camera -> rectify -> apriltag graphs use; here
they are switched with `rectified_topics:=true` (default false = the stand-alone file's behaviour)."""
import launch
from launch.actions import DeclareLaunchArgument
from launch.conditions import IfCondition, UnlessCondition
from launch.substitutions import LaunchConfiguration
from launch_ros.actions import ComposableNodeContainer
from launch_ros.descriptions import ComposableNode


def generate_launch_description():
    args = [
        DeclareLaunchArgument('size', default_value='0.22', description='tag edge, metres'),
        DeclareLaunchArgument('max_tags', default_value='64'),
        DeclareLaunchArgument('tile_size', default_value='4'),
        DeclareLaunchArgument('tag_family', default_value='tag36h11'),
        DeclareLaunchArgument('backends', default_value='CUDA',
                              description="'CUDA' = cuAprilTags-compatible mode (tag36h11 only); any other list = VPI-compatible mode"),
        DeclareLaunchArgument('rectified_topics', default_value='false',
                              description='remap image / camera_info onto image_rect / camera_info_rect as the core fragment does'),
    ]
    params = [{'size': LaunchConfiguration('size'), 'max_tags': LaunchConfiguration('max_tags'),
               'tile_size': LaunchConfiguration('tile_size'), 'tag_family': LaunchConfiguration('tag_family'),
               'backends': LaunchConfiguration('backends')}]

    def make_node(remappings):
        return ComposableNode(package='isaac_ros_apriltag', plugin='nvidia::isaac_ros::apriltag::AprilTagNode', name='apriltag',
                              parameters=params, remappings=remappings)

    def make_container(node, condition):
        return ComposableNodeContainer(
            package='rclcpp_components', name='apriltag_container', namespace='', executable='component_container_mt',
            composable_node_descriptions=[node], output='screen', condition=condition)
    plain = make_container(make_node([]), UnlessCondition(LaunchConfiguration('rectified_topics')))
    rect = make_container(make_node([('image', 'image_rect'), ('camera_info', 'camera_info_rect')]),
                          IfCondition(LaunchConfiguration('rectified_topics')))
    return launch.LaunchDescription(args + [plain, rect])
