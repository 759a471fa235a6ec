// stand-in for <pcl/common/transforms.h> (test infrastructure, see ref_stubs.h)
#pragma once
#include "../../ref_stubs.h"
