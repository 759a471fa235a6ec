// stand-in for <pcl/common/centroid.h> (test infrastructure, see ref_stubs.h)
#pragma once
#include "../../ref_stubs.h"
