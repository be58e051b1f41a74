// TEST INFRASTRUCTURE (tests/hipemu/build_ref_host.py only): lets the host compiler take the REFERENCE's CUDA sources where they lie
// under /root/reference, on top of tests/hipemu's stand-in HIP runtime (fibers).  Never included by the product.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdlib>
#include <stdexcept>
typedef hipError_t cudaError_t;
#define cudaMemcpy hipMemcpy
#define cudaMemcpyDeviceToHost hipMemcpyDeviceToHost
#define cudaMemset hipMemset
#define cudaMalloc hipMalloc
#define cudaFree hipFree
#define cudaDeviceSynchronize hipDeviceSynchronize
#define cudaSuccess hipSuccess
#define cudaGetErrorString hipGetErrorString
#define __trap abort
// what CUDA's headers give device code and a plain host compiler does not: float overloads of the C math functions in the global
// namespace (exp(float) IS expf in the reference's kernels), mixed float / double min / max (resolved in double, as CUDA does),
// double3, atomicAdd on a float with a double operand (converted, as CUDA's implicit conversion does)
#include <math.h>
struct double3 { double x, y, z; };
struct alignas(32) double4 { double x, y, z, w; };
inline double max(double a, float b) { return fmax(a, (double)b); }
inline double max(float a, double b) { return fmax((double)a, b); }
inline double min(double a, float b) { return fmin(a, (double)b); }
inline double min(float a, double b) { return fmin((double)a, b); }
inline float atomicAdd(float* p, double v) { return atomicAdd(p, (float)v); }
