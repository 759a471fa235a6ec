// stand-in for <tf/transform_broadcaster.h> (test infrastructure, see ref_stubs.h)
#pragma once
#include "../ref_stubs.h"
