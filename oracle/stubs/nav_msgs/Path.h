// stand-in for <nav_msgs/Path.h> (test infrastructure, see ref_stubs.h)
#pragma once
#include "../ref_stubs.h"
