// penta_apply.h — solves H X = R for MANY right-hand sides with the block LDL^T factors that
// penta_ldl_kernel (one workgroup, one-sided) left in HBM: one wavefront per right-hand side,
// as many workgroups as needed, no synchronisation between them.  This is the path of
// CalcLagrangeMultipliers (reference optimizer/trajectory_optimizer.cc:1371-1396: H^-1 J^T, one
// column per equality constraint, 120-240 columns for the example models): the factorisation is
// a serial chain on one CU, the substitutions are embarrassingly parallel over columns.
//
// Stored factors per block row i (row-major, row stride ks, written by penta_ldl_kernel):
//   Ust_i = D^-1 U_i (strict upper; = L_i^T), Hst_i = D^-1 Ht_i, Est_i = D^-1 Et_i, Dst_i = 1/d.
// Forward  (i ascending):  rt_i = L_i^-1 (r_i - Ht_{i-1}^T Dn rt_{i-1} - Et_{i-2}^T Dn rt_{i-2})
// Backward (i descending): x_i  = (D^-1 U_i)^-1 (D^-1 rt_i - D^-1 Ht_i x_{i+1} - D^-1 Et_i x_{i+2})
// Both run in "push" form: the triangular solve produces one component per step (v_readlane),
// and each component is at once pushed into the pending right-hand sides of the next two block
// rows - the mat-vec products ride in the latency shadow of the dependent chain.
//   forward, lane c:  element [jj][c] of the three blocks (a column: coalesced 8-byte loads)
//   backward, lane r: row r of the three blocks (16-byte loads)
#pragma once

#include <hip/hip_runtime.h>

#include "penta_ldl.h"

namespace idto_dev {

template <int K>
__global__ void __launch_bounds__(256)
penta_apply_kernel(int n, int k, const double* __restrict__ Ust, const double* __restrict__ Hst,
                   const double* __restrict__ Est, const double* __restrict__ Dst, const double* __restrict__ rhs,
                   double rhs_sign, int nrhs, double* __restrict__ x) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.x * (blockDim.x >> 6) + wave;
  if (j >= nrhs) return;  // no barriers below
  constexpr int ks = ldl_ks(K), KS2 = K * ks, KP = (K + 1) / 2;
  const size_t nk = (size_t)n * k;
  const int c = (lane < K) ? lane : K - 1;  // lanes >= K shadow lane K-1 (never stored)
  const bool live = lane < K;
  double* rtw = lds + (size_t)wave * n * K;  // rt_i[c] of this right-hand side, all rows

  // ---- forward substitution
  double pend1 = 0.0, pend2 = 0.0;  // pushed into rows i+1 (pend1) and i+2 (pend2)
  for (int i = 0; i < n; ++i) {
    const double* U = Ust + (size_t)i * KS2 + c;
    const double* Hh = Hst + (size_t)i * KS2 + c;
    const double* Ee = Est + (size_t)i * KS2 + c;
    double u[K], h[K], e[K];
#pragma unroll
    for (int jj = 0; jj < K; ++jj) { u[jj] = U[jj * ks]; h[jj] = Hh[jj * ks]; e[jj] = Ee[jj * ks]; }
    double v = ((c < k) ? rhs_sign * rhs[(size_t)j * nk + (size_t)i * k + c] : 0.0) + pend1;
    double a1 = pend2, a2 = 0.0;
#pragma unroll
    for (int jj = 0; jj < K; ++jj) {
      const double t = rdlane(v, jj);       // rt_i[jj]: final once the steps before it are done
      v = __builtin_fma(-u[jj], t, v);      // L[c][jj] = (D^-1 U)[jj][c], zero for c <= jj
      a1 = __builtin_fma(-h[jj], t, a1);    // (Ht_i^T Dn rt_i)[c]
      a2 = __builtin_fma(-e[jj], t, a2);    // (Et_i^T Dn rt_i)[c]
    }
    if (live) rtw[i * K + lane] = v;
    pend1 = a1;
    pend2 = a2;
  }

  // ---- back substitution
  // v: pending right-hand side of the row being solved; p: what has been pushed so far into the
  // row after it (-(D^-1 Et_{i-1}) x_{i+1}, pushed one iteration earlier)
  const int r = c;
  double v = Dst[(size_t)(n - 1) * K + r] * rtw[(n - 1) * K + r];
  double p = 0.0;
  for (int i = n - 1; i >= 0; --i) {
    const double2* pu = reinterpret_cast<const double2*>(Ust + (size_t)i * KS2 + r * ks);
    const double2* ph = reinterpret_cast<const double2*>(Hst + (size_t)(i > 0 ? i - 1 : 0) * KS2 + r * ks);
    const double2* pe = reinterpret_cast<const double2*>(Est + (size_t)(i > 1 ? i - 2 : 0) * KS2 + r * ks);
    double2 U2[KP], H2[KP], E2[KP];
#pragma unroll
    for (int m = KP - 1; m >= 0; --m) { U2[m] = pu[m]; H2[m] = ph[m]; E2[m] = pe[m]; }
    const double hs = (i > 0) ? -1.0 : 0.0, es = (i > 1) ? -1.0 : 0.0;  // rows -1, -2 do not exist
    const double next_rt = (i > 0) ? Dst[(size_t)(i - 1) * K + r] * rtw[(i - 1) * K + r] : 0.0;
    double pn = 0.0;
#pragma unroll
    for (int jj = K - 1; jj >= 0; --jj) {
      const double xj = rdlane(v, jj);
      const double ujj = (jj & 1) ? U2[jj / 2].y : U2[jj / 2].x;
      const double hjj = (jj & 1) ? H2[jj / 2].y : H2[jj / 2].x;
      const double ejj = (jj & 1) ? E2[jj / 2].y : E2[jj / 2].x;
      v = __builtin_fma(-ujj, xj, v);       // strictly upper: rows >= jj keep their value
      p = __builtin_fma(hjj * hs, xj, p);   // row i-1: -(D^-1 Ht_{i-1}) x_i
      pn = __builtin_fma(ejj * es, xj, pn); // row i-2: -(D^-1 Et_{i-2}) x_i
    }
    if (live && lane < k) x[(size_t)j * nk + (size_t)i * k + lane] = v;
    v = next_rt + p;
    p = pn;
  }
}

}  // namespace idto_dev
