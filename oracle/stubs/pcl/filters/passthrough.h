// stand-in for <pcl/filters/passthrough.h> (test infrastructure, see ref_stubs.h)
#pragma once
#include "../../ref_stubs.h"
