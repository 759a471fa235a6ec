// stand-in for <pcl/PCLPointCloud2.h> (test infrastructure, see ref_stubs.h)
#pragma once
#include "../ref_stubs.h"
