"""Loads the AprilTag component into a multi-threaded container -- plugin identity, node name, container type and
remappable topics as in the reference's launch file (launch/isaac_ros_apriltag.launch.py:25-40)."""
import launch
from launch.actions import DeclareLaunchArgument
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
    ]
    node = ComposableNode(
        package='isaac_ros_apriltag',
        plugin='nvidia::isaac_ros::apriltag::AprilTagNode',
        name='apriltag',
        parameters=[{'size': LaunchConfiguration('size'), 'max_tags': LaunchConfiguration('max_tags'),
                     'tile_size': LaunchConfiguration('tile_size'), 'tag_family': LaunchConfiguration('tag_family'),
                     'backends': LaunchConfiguration('backends')}],
        remappings=[('image', 'image_rect'), ('camera_info', 'camera_info_rect')])
    container = ComposableNodeContainer(
        package='rclcpp_components', name='apriltag_container', namespace='',
        executable='component_container_mt', composable_node_descriptions=[node], output='screen')
    return launch.LaunchDescription(args + [container])
