// Stand-in for <cuda_runtime.h> (see README.md): HIP runtime + the handful of cuda* names the
// reference's orchestrator uses (aggregator_impl.cu:197,226; auxiliary.h:22-29).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <stdexcept>
#define cudaMemcpy hipMemcpy
#define cudaMemset hipMemset
#define cudaMemcpyDeviceToHost hipMemcpyDeviceToHost
#define cudaDeviceSynchronize hipDeviceSynchronize
#define cudaSuccess hipSuccess
#define cudaGetErrorString hipGetErrorString
// CUDA's math headers provide mixed signed/unsigned min/max (auxiliary.h:8-20 calls
// min(unsigned, int)); CUDA converts the int to unsigned.
__device__ __forceinline__ unsigned int min(unsigned int a, int b) { unsigned int ub = (unsigned int)b; return a < ub ? a : ub; }
__device__ __forceinline__ unsigned int min(int a, unsigned int b) { unsigned int ua = (unsigned int)a; return ua < b ? ua : b; }
