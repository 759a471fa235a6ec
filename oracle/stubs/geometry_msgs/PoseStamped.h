// stand-in for <geometry_msgs/PoseStamped.h> (test infrastructure, see ref_stubs.h)
#pragma once
#include "../ref_stubs.h"
