// TEST INFRASTRUCTURE: stand-in for <ATen/cuda/CUDAContext.h>: the one name the reference's launch wrappers use.
#pragma once
#include <cuda_runtime.h>
namespace at { namespace cuda {
static inline cudaStream_t getCurrentCUDAStream() { return nullptr; }
}}
