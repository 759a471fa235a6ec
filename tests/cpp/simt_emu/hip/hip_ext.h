// (hipExtLaunchKernelGGL lives in the stand-in for hip_runtime.h)
#include "hip_runtime.h"
