#pragma once  // float atomicAdd is native in HIP
#include <hip/hip_runtime.h>
