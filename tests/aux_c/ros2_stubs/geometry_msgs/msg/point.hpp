#pragma once
#include "geometry_msgs/msg/types.hpp"
