// stand-in for <pcl/point_cloud.h> (test infrastructure, see ref_stubs.h)
#pragma once
#include "../ref_stubs.h"
