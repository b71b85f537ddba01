// Host build of sam6d_amd/csrc/s6d_rot.h for CPU-side logic tests (tests/test_rot_host.py).
#include "../../sam6d_amd/csrc/s6d_rot.h"
extern "C" void rot_from_h_host(const double *H, int n, double *R) {
  for (int i = 0; i < n; ++i) s6d::rot_from_h(H + 9 * i, R + 9 * i);
}
