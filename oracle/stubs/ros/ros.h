// stand-in for <ros/ros.h> (test infrastructure, see ref_stubs.h)
#pragma once
#include "../ref_stubs.h"
