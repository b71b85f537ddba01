// TEST INFRASTRUCTURE: stand-in for <ATen/ATen.h>; the *_gpu.cu files of the reference take raw pointers and need nothing from it.
#pragma once
