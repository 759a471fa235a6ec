// stand-in for <pcl/visualization/pcl_visualizer.h> (test infrastructure, see ref_stubs.h)
#pragma once
#include "../../ref_stubs.h"
// The reference's sources use unqualified cout / endl / setw without a using-directive of their own
// (erasor_utils.cpp:141, OfflineMapUpdater.cpp:257,461): in its real environment a third-party header
// reached from here leaks <iomanip> and `using namespace std`.  Reproduced so the sources compile unmodified.
#include <iomanip>
using namespace std;
