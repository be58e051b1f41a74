// TEST INFRASTRUCTURE (oracle/_ref build only): lets hipcc compile the reference's CUDA sources
// where they lie under /root/reference.  Never included by the product.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdlib>
#define cudaMemcpy hipMemcpy
#define cudaMemcpyDeviceToHost hipMemcpyDeviceToHost
#define cudaMemset hipMemset
#define cudaMalloc hipMalloc
#define cudaFree hipFree
#define cudaDeviceSynchronize hipDeviceSynchronize
#define cudaSuccess hipSuccess
#define cudaGetErrorString hipGetErrorString
#define __trap abort
