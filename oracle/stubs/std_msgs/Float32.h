// stand-in for <std_msgs/Float32.h> (test infrastructure, see ref_stubs.h)
#pragma once
#include "../ref_stubs.h"
