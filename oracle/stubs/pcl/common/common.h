// stand-in for <pcl/common/common.h> (test infrastructure, see ref_stubs.h)
#pragma once
#include "../../ref_stubs.h"
