#pragma once
#include <string>
#include "builtin_interfaces/msg/time.hpp"
namespace std_msgs { namespace msg { struct Header { builtin_interfaces::msg::Time stamp; std::string frame_id; }; } }
