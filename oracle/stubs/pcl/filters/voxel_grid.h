// stand-in for <pcl/filters/voxel_grid.h> (test infrastructure, see ref_stubs.h)
#pragma once
#include "../../ref_stubs.h"
