// TEST INFRASTRUCTURE.  Stand-in for <cuda_runtime.h> used ONLY by oracle/build_ref.py to compile the reference's PointNet++
// CUDA kernels (SAM-6D/Pose_Estimation_Model/model/pointnet2/_ext_src/src/*_gpu.cu, read in place from /root/reference) for
// the HOST on top of the emulated runtime of tests/host_cc/hipemu: blocks run one after another, the threads of a block are
// fibers, __syncthreads is a rendezvous, __shared__ is a static.  The kernels use nothing beyond threadIdx / blockIdx /
// blockDim / gridDim / __shared__ / __syncthreads / atomicAdd / min / max, all of which that header provides.
#pragma once
#include <hip/hip_runtime.h>

typedef hipStream_t cudaStream_t;
typedef hipError_t cudaError_t;
enum { cudaSuccess = hipSuccess };
static inline cudaError_t cudaGetLastError() { return cudaSuccess; }
static inline const char *cudaGetErrorString(cudaError_t) { return "pn2_ref: no error"; }
