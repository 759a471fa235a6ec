// stand-in for <jsk_recognition_msgs/PolygonArray.h> (test infrastructure, see ref_stubs.h)
#pragma once
#include "../ref_stubs.h"
