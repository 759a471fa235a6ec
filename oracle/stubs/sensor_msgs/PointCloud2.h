// stand-in for <sensor_msgs/PointCloud2.h> (test infrastructure, see ref_stubs.h)
#pragma once
#include "../ref_stubs.h"
