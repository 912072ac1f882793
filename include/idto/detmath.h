// detmath.h — deterministic elementary functions (sin/cos, exp, log) built only
// from IEEE-754 +,-,*,/ and bit manipulation, so that the host (g++) and the
// gfx950 device (hipcc) produce the SAME BITS for the same argument when both
// are compiled with -ffp-contract=off.
//
// Why: the hot path differentiates inverse dynamics by forward differences with
// dq ~ 1.5e-8 (reference optimizer/trajectory_optimizer.cc:504-511), which
// amplifies any 1-ulp host/device disagreement in tau by 1/dq ~ 6.7e7.  glibc
// and the ROCm device libm do not agree to the last bit, these do.
//
// Algorithms: Cody-Waite range reduction + the classic fdlibm polynomial
// kernels (public constants).  Accuracy is ~1 ulp on the domains used here
// (|x| < 1e5 for sin/cos, |x| < 700 for exp), checked against libm in
// tests/test_detmath.py.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define IDTO_HD __host__ __device__ __forceinline__
#else
#define IDTO_HD inline
#endif

namespace idto {
namespace detmath {

IDTO_HD double from_bits(uint64_t u) {
  double d;
  __builtin_memcpy(&d, &u, sizeof(d));
  return d;
}
IDTO_HD uint64_t to_bits(double d) {
  uint64_t u;
  __builtin_memcpy(&u, &d, sizeof(u));
  return u;
}

// 2^k for -1022 <= k <= 1023.
IDTO_HD double pow2i(int k) { return from_bits((uint64_t)(k + 1023) << 52); }

// sin and cos of x (|x| up to ~1e5 keeps full accuracy).
IDTO_HD void sincos(double x, double* s_out, double* c_out) {
  const double invpio2 = 6.36619772367581382433e-01;
  const double pio2_1 = 1.57079632673412561417e+00;  // first 33 bits of pi/2
  const double pio2_2 = 6.07710050630396597660e-11;  // next 33 bits
  const double pio2_3 = 2.02226624871116645580e-21;  // next 33 bits
  const double pio2_3t = 8.47842766036889956997e-32;
  const double fn = __builtin_rint(x * invpio2);
  double r = x - fn * pio2_1;
  r = r - fn * pio2_2;
  r = r - fn * pio2_3;
  r = r - fn * pio2_3t;
  const int n = ((int)(long long)fn) & 3;

  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double z = r * r;
  const double v = z * r;
  const double rs = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  const double sn = r + v * (S1 + z * rs);
  const double rc = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  const double cs = 1.0 - (0.5 * z - z * rc);
  // quadrant n: (sn, cs), (cs, -sn), (-sn, -cs), (-cs, sn) - as selects, so that the device code has no branch here
  // and the sines and cosines of several joints can be evaluated side by side
  const bool swap = (n & 1) != 0;
  const double sv = swap ? cs : sn, cv = swap ? sn : cs;
  *s_out = (n & 2) ? -sv : sv;
  *c_out = ((n + 1) & 2) ? -cv : cv;
}

IDTO_HD double exp(double x) {
  // (the special cases are selects at the end: no branch in the device code; the polynomial runs on a clamped
  // argument so that the integer conversions below stay defined)
  const bool isnan = x != x, over = x > 709.78, under = x < -745.2;
  const double xc = (isnan || over || under) ? 0.0 : x;
  const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
               invln2 = 1.44269504088896338700e+00;
  const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
               P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06,
               P5 = 4.13813679705723846039e-08;
  const double fk = __builtin_rint(xc * invln2);
  const int k = (int)fk;
  const double hi = xc - fk * ln2HI;
  const double lo = fk * ln2LO;
  const double r = hi - lo;
  const double t = r * r;
  const double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
  const double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
  const int k1 = k / 2, k2 = k - k1;
  const double res = (y * pow2i(k1)) * pow2i(k2);
  return isnan ? x : (over ? from_bits(0x7ff0000000000000ull) : (under ? 0.0 : res));
}

// Natural logarithm for finite x > 0 (subnormals included).
IDTO_HD double log(double x0) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
               Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
               Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  const uint64_t u0 = to_bits(x0);
  const bool isnan = x0 != x0, neg = x0 < 0.0, zero = x0 == 0.0, inf = u0 == 0x7ff0000000000000ull;
  const bool sub = (u0 >> 52) == 0;               // subnormal: scale up by 2^54
  const double x = sub ? x0 * 18014398509481984.0 : x0;
  const uint64_t u = to_bits(x);
  int k = sub ? -54 : 0;
  uint32_t hx = (uint32_t)(u >> 32);
  const uint32_t lx = (uint32_t)u;
  k += (int)(hx >> 20) - 1023;
  hx &= 0x000fffffu;
  const uint32_t i = (hx + 0x95f64u) & 0x100000u;
  const double m = from_bits(((uint64_t)(hx | (i ^ 0x3ff00000u)) << 32) | lx);  // in [sqrt(2)/2, sqrt(2))
  k += (int)(i >> 20);
  const double f = m - 1.0;
  const double s = f / (2.0 + f);
  const double dk = (double)k;
  const double z = s * s;
  const double w = z * z;
  const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  const double res = dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
  // special cases in the order the branches had: NaN, negative, zero, +inf
  return isnan ? x0 : (neg ? from_bits(0x7ff8000000000000ull) : (zero ? from_bits(0xfff0000000000000ull) : (inf ? x0 : res)));
}

}  // namespace detmath
}  // namespace idto
