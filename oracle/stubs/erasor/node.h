// stand-in for <erasor/node.h> (test infrastructure, see ref_stubs.h)
#pragma once
#include "../ref_stubs.h"
