// stand-in for <std_msgs/Int32.h> (test infrastructure, see ref_stubs.h)
#pragma once
#include "../ref_stubs.h"
