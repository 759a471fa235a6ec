// stand-in for <pcl/kdtree/kdtree_flann.h> (test infrastructure, see ref_stubs.h)
#pragma once
#include "../../ref_stubs.h"
