/* Compile-time shim (TEST INFRASTRUCTURE ONLY): lets the reference's CUDA sources compile UNMODIFIED, where they lie
 * under /root/reference, with hipcc for gfx950 so the real reference kernels can serve as a GPU-side oracle
 * (oracle/build_ref.py).  Nothing here ships in the product library. */
#pragma once
#include <hip/hip_runtime.h>
#include <cfloat>
#define cudaError_t hipError_t
#define cudaSuccess hipSuccess
#define cudaGetErrorString hipGetErrorString
#define cudaDeviceSynchronize hipDeviceSynchronize
#define cudaMemcpy hipMemcpy
#define cudaMemset hipMemset
#define cudaMemcpyDeviceToHost hipMemcpyDeviceToHost
#define cudaMemcpyHostToDevice hipMemcpyHostToDevice
#define cudaMalloc hipMalloc
#define cudaFree hipFree
#define cudaEvent_t hipEvent_t
#define cudaEventCreate hipEventCreate
#define cudaEventRecord hipEventRecord
#define cudaEventSynchronize hipEventSynchronize
#define cudaEventElapsedTime hipEventElapsedTime
#ifndef __trap
#define __trap() __builtin_trap()
#endif
