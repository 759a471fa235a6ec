// stand-in for <geometry_msgs/Polygon.h> (test infrastructure, see ref_stubs.h)
#pragma once
#include "../ref_stubs.h"
