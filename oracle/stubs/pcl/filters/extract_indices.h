// stand-in for <pcl/filters/extract_indices.h> (test infrastructure, see ref_stubs.h)
#pragma once
#include "../../ref_stubs.h"
