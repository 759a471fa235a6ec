// stand-in for <pcl_conversions/pcl_conversions.h> (test infrastructure, see ref_stubs.h)
#pragma once
#include "../ref_stubs.h"
