// TEST INFRASTRUCTURE: stand-in for <cuda.h> (see cuda_runtime.h in this directory).
