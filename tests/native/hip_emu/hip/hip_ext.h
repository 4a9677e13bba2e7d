// host stand-in: hipExtLaunchKernelGGL lives in hip_runtime.h here
#pragma once
#include <hip/hip_runtime.h>
