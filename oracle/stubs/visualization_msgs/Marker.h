// stand-in for <visualization_msgs/Marker.h> (test infrastructure, see ref_stubs.h)
#pragma once
#include "../ref_stubs.h"
