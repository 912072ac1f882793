// dev_math.h — small fixed-size vector algebra for the gfx950 kernels.
//
// Every expression is written with an explicit association order; together with
// -ffp-contract=off this makes each function produce the same bits as the same
// formula evaluated on the host.  The products of 3-vectors and 3x3 matrices
// (dot, cross, M v, M^T v, M M) use EXPLICIT fused multiply-adds in a fixed form, identically in
// oracle/rigid_body.h: fma is one correctly rounded IEEE operation on both sides, and these five
// primitives are ~60 % of the arithmetic of an inverse-dynamics evaluation.
#pragma once

#include <hip/hip_runtime.h>

#include "idto/detmath.h"

#define IDTO_DEV __device__ __forceinline__

namespace idto_dev {

struct V3 {
  double x, y, z;
};
IDTO_DEV V3 mk(double x, double y, double z) { V3 v; v.x = x; v.y = y; v.z = z; return v; }
IDTO_DEV V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
IDTO_DEV V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
IDTO_DEV V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
IDTO_DEV V3 operator*(V3 a, double s) { return mk(a.x * s, a.y * s, a.z * s); }
IDTO_DEV V3 operator/(V3 a, double s) { return mk(a.x / s, a.y / s, a.z / s); }
IDTO_DEV double fma3(double a0, double b0, double a1, double b1, double a2, double b2) {
  return __builtin_fma(a2, b2, __builtin_fma(a1, b1, a0 * b0));  // (a0 b0 (+) a1 b1) (+) a2 b2, two fused steps
}
IDTO_DEV double fms(double a, double b, double c, double d) { return __builtin_fma(a, b, -(c * d)); }  // a b - c d
IDTO_DEV double dot(V3 a, V3 b) { return fma3(a.x, b.x, a.y, b.y, a.z, b.z); }
IDTO_DEV V3 cross(V3 a, V3 b) {
  return mk(fms(a.y, b.z, a.z, b.y), fms(a.z, b.x, a.x, b.z), fms(a.x, b.y, a.y, b.x));
}
IDTO_DEV V3 ldv3(const double* p) { return mk(p[0], p[1], p[2]); }

struct M3 {
  double m[9];  // row-major
};
IDTO_DEV M3 ident3() {
  M3 R;
  R.m[0] = 1; R.m[1] = 0; R.m[2] = 0; R.m[3] = 0; R.m[4] = 1; R.m[5] = 0; R.m[6] = 0; R.m[7] = 0; R.m[8] = 1;
  return R;
}
IDTO_DEV M3 ldm3(const double* p) {
  M3 R;
#pragma unroll
  for (int i = 0; i < 9; ++i) R.m[i] = p[i];
  return R;
}
IDTO_DEV V3 operator*(const M3& R, V3 v) {
  return mk(fma3(R.m[0], v.x, R.m[1], v.y, R.m[2], v.z), fma3(R.m[3], v.x, R.m[4], v.y, R.m[5], v.z),
            fma3(R.m[6], v.x, R.m[7], v.y, R.m[8], v.z));
}
IDTO_DEV V3 tmul(const M3& R, V3 v) {  // R^T v
  return mk(fma3(R.m[0], v.x, R.m[3], v.y, R.m[6], v.z), fma3(R.m[1], v.x, R.m[4], v.y, R.m[7], v.z),
            fma3(R.m[2], v.x, R.m[5], v.y, R.m[8], v.z));
}
IDTO_DEV M3 operator*(const M3& A, const M3& B) {
  M3 C;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c)
      C.m[3 * r + c] = fma3(A.m[3 * r], B.m[c], A.m[3 * r + 1], B.m[3 + c], A.m[3 * r + 2], B.m[6 + c]);
  return C;
}
IDTO_DEV V3 col(const M3& R, int c) { return mk(R.m[c], R.m[3 + c], R.m[6 + c]); }

// Rotation by the angle with sine s and cosine c about the unit axis a.
IDTO_DEV M3 axis_angle(V3 a, double s, double c) {
  const V3 sa = a * s;
  const V3 ca = a * (1.0 - c);
  M3 R;
  double t;
  t = ca.x * a.y; R.m[1] = t - sa.z; R.m[3] = t + sa.z;
  t = ca.x * a.z; R.m[2] = t + sa.y; R.m[6] = t - sa.y;
  t = ca.y * a.z; R.m[5] = t - sa.x; R.m[7] = t + sa.x;
  R.m[0] = ca.x * a.x + c;
  R.m[4] = ca.y * a.y + c;
  R.m[8] = ca.z * a.z + c;
  return R;
}

// Rotation matrix of a (possibly un-normalised) quaternion [w x y z].
IDTO_DEV M3 quat_to_rot(const double* q) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double n2 = ((w * w + x * x) + y * y) + z * z;
  const double s2 = 2.0 / n2;
  const double sx = s2 * x, sy = s2 * y, sz = s2 * z;
  const double swx = sx * w, swy = sy * w, swz = sz * w;
  const double sxx = sx * x, sxy = sy * x, sxz = sz * x;
  const double syy = sy * y, syz = sz * y, szz = sz * z;
  M3 R;
  R.m[0] = (1.0 - syy) - szz; R.m[1] = sxy - swz;         R.m[2] = sxz + swy;
  R.m[3] = sxy + swz;         R.m[4] = (1.0 - sxx) - szz; R.m[5] = syz - swx;
  R.m[6] = sxz - swy;         R.m[7] = syz + swx;         R.m[8] = (1.0 - sxx) - syy;
  return R;
}

// x + (value of x in lane ^ mask): one step of the butterfly sum over the lanes
// that cooperate on one inverse-dynamics evaluation.
IDTO_DEV double xor_add(double x, int mask) { return x + __shfl_xor(x, mask); }
IDTO_DEV V3 tree_sum(V3 v, int npaths) {
  for (int stride = 1; stride < npaths; stride *= 2) {
    v.x = xor_add(v.x, stride);
    v.y = xor_add(v.y, stride);
    v.z = xor_add(v.z, stride);
  }
  return v;
}

}  // namespace idto_dev
