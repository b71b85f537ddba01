// 3x3 rotation from a cross-covariance matrix; shared by the pose kernels.  Plain C++ so the same
// source is also compiled for the host by tests/test_rot_host.py (logic check without a GPU).
#pragma once
#include <math.h>
#ifndef S6D_HD
#ifdef __HIPCC__
#define S6D_HD __host__ __device__
#else
#define S6D_HD
#endif
#endif

namespace s6d {

// R = V diag(1,1,det(V U^T)) U^T for H = U S V^T  (model_utils.py:343-347), computed from the
// eigenvectors of H^T H:  v1,v2 (two largest), u_i = H v_i / |H v_i|, third axes by cross
// products -- which yields exactly the determinant-corrected product.
S6D_HD inline void rot_from_h(const double H[9], double R[9]) {
  // A = H^T H (symmetric)
  double a00 = H[0] * H[0] + H[3] * H[3] + H[6] * H[6];
  double a01 = H[0] * H[1] + H[3] * H[4] + H[6] * H[7];
  double a02 = H[0] * H[2] + H[3] * H[5] + H[6] * H[8];
  double a11 = H[1] * H[1] + H[4] * H[4] + H[7] * H[7];
  double a12 = H[1] * H[2] + H[4] * H[5] + H[7] * H[8];
  double a22 = H[2] * H[2] + H[5] * H[5] + H[8] * H[8];
  double V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};  // columns = eigenvectors
  const double tiny = 1e-300;
#pragma unroll 1
  for (int sweep = 0; sweep < 10; ++sweep) {
    const double off = fabs(a01) + fabs(a02) + fabs(a12);
    if (off <= 1e-30 * (fabs(a00) + fabs(a11) + fabs(a22)) + tiny) break;
    // (p,q) = (0,1)
    if (fabs(a01) > tiny) {
      const double th = (a11 - a00) / (2.0 * a01);
      const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
      const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
      const double n00 = a00 - t * a01, n11 = a11 + t * a01;
      const double n02 = c * a02 - s * a12, n12 = s * a02 + c * a12;
      a00 = n00; a11 = n11; a01 = 0; a02 = n02; a12 = n12;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const double vp = V[r * 3 + 0], vq = V[r * 3 + 1];
        V[r * 3 + 0] = c * vp - s * vq;
        V[r * 3 + 1] = s * vp + c * vq;
      }
    }
    // (0,2)
    if (fabs(a02) > tiny) {
      const double th = (a22 - a00) / (2.0 * a02);
      const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
      const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
      const double n00 = a00 - t * a02, n22 = a22 + t * a02;
      const double n01 = c * a01 - s * a12, n12 = s * a01 + c * a12;
      a00 = n00; a22 = n22; a02 = 0; a01 = n01; a12 = n12;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const double vp = V[r * 3 + 0], vq = V[r * 3 + 2];
        V[r * 3 + 0] = c * vp - s * vq;
        V[r * 3 + 2] = s * vp + c * vq;
      }
    }
    // (1,2)
    if (fabs(a12) > tiny) {
      const double th = (a22 - a11) / (2.0 * a12);
      const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
      const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
      const double n11 = a11 - t * a12, n22 = a22 + t * a12;
      const double n01 = c * a01 - s * a02, n02 = s * a01 + c * a02;
      a11 = n11; a22 = n22; a12 = 0; a01 = n01; a02 = n02;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const double vp = V[r * 3 + 1], vq = V[r * 3 + 2];
        V[r * 3 + 1] = c * vp - s * vq;
        V[r * 3 + 2] = s * vp + c * vq;
      }
    }
  }
  // order: i1 = largest eigenvalue, i2 = second
  double l[3] = {a00, a11, a22};
  int i1 = 0;
  if (l[1] > l[i1]) i1 = 1;
  if (l[2] > l[i1]) i1 = 2;
  int i2 = (i1 + 1) % 3, i3 = (i1 + 2) % 3;
  if (l[i3] > l[i2]) { const int tmp = i2; i2 = i3; i3 = tmp; }
  double v1[3] = {V[0 + i1], V[3 + i1], V[6 + i1]};
  double v2[3] = {V[0 + i2], V[3 + i2], V[6 + i2]};
  double u1[3], u2[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    u1[r] = H[r * 3 + 0] * v1[0] + H[r * 3 + 1] * v1[1] + H[r * 3 + 2] * v1[2];
    u2[r] = H[r * 3 + 0] * v2[0] + H[r * 3 + 1] * v2[1] + H[r * 3 + 2] * v2[2];
  }
  double n1 = sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
  if (!(n1 > tiny)) {  // H == 0: SVD of zero gives U = V = I
    R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
    return;
  }
  u1[0] /= n1; u1[1] /= n1; u1[2] /= n1;
  const double d = u1[0] * u2[0] + u1[1] * u2[1] + u1[2] * u2[2];
  u2[0] -= d * u1[0]; u2[1] -= d * u1[1]; u2[2] -= d * u1[2];
  double n2 = sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
  if (!(n2 > 1e-14 * n1)) {  // rank one: any unit vector orthogonal to u1 (the reference result is arbitrary too)
    const double ax = fabs(u1[0]) < 0.9 ? 1.0 : 0.0, ay = 1.0 - ax;
    u2[0] = u1[1] * 0.0 - u1[2] * ay; u2[1] = u1[2] * ax - u1[0] * 0.0; u2[2] = u1[0] * ay - u1[1] * ax;
    n2 = sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
  }
  u2[0] /= n2; u2[1] /= n2; u2[2] /= n2;
  const double u3[3] = {u1[1] * u2[2] - u1[2] * u2[1], u1[2] * u2[0] - u1[0] * u2[2], u1[0] * u2[1] - u1[1] * u2[0]};
  const double v3[3] = {v1[1] * v2[2] - v1[2] * v2[1], v1[2] * v2[0] - v1[0] * v2[2], v1[0] * v2[1] - v1[1] * v2[0]};
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) R[r * 3 + c] = v1[r] * u1[c] + v2[r] * u2[c] + v3[r] * u3[c];
}

}  // namespace s6d
