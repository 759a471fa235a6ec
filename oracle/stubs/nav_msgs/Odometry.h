// stand-in for <nav_msgs/Odometry.h> (test infrastructure, see ref_stubs.h)
#pragma once
#include "../ref_stubs.h"
