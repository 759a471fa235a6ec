// stand-in for <pcl/io/pcd_io.h> (test infrastructure, see ref_stubs.h)
#pragma once
#include "../../ref_stubs.h"
