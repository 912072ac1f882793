// penta_apply.h — solves H X = R for MANY right-hand sides with the block LDL^T factors that
// penta_ldl_kernel (one- or two-sided) left in HBM: one wavefront per right-hand side,
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

#include "batch.h"
#include "penta_ldl.h"

namespace idto_dev {

// Right-hand sides that are not stored: column 0 = g, column 1 + r = row r of the equality constraints' Jacobian
// J[(t, dof), :] = [dtau_t/dq_{t-1} (t > 1) | dtau_t/dq_t (t > 0) | dtau_t/dq_{t+1}](dof, :), i.e. rows of the slab
// records (constraints.h; reference optimizer/trajectory_optimizer.cc:1292-1334).  The substitution kernel reads
// them where they are instead of from (n_eq + 1) staged columns (4 MB for the allegro hand at N = 60).
// slab == nullptr: the right-hand sides are the array `rhs`.
struct RhsSource {
  const double* slab;
  int slab_stride, nu, nv, r0;   // r0: block rows the solved sub-system skips at the top (their x is rhs = 0 here)
  const int* dofs;
  const double* g;
  AltSel alt;
};

// Column-major copies of the three factor blocks of every row (grid (n, 3)): the forward pass
// reads COLUMNS of the row-major blocks; from the transposed copy a lane gets its column with
// 16-byte loads, half as many load instructions on the dependent chain.
template <int K>
__global__ void penta_factor_transpose_kernel(const double* __restrict__ Ust, const double* __restrict__ Hst,
                                              const double* __restrict__ Est, double* __restrict__ T) {
  constexpr int ks = ldl_ks(K), KS2 = K * ks;
  const int i = blockIdx.x, which = blockIdx.y, n = gridDim.x;
  const double* src = (which == 0 ? Ust : which == 1 ? Hst : Est) + (size_t)i * KS2;
  double* dst = T + ((size_t)which * n + i) * KS2;
  for (int idx = threadIdx.x; idx < KS2; idx += blockDim.x) {
    const int c = idx / ks, jj = idx - c * ks;
    dst[idx] = (jj < K) ? src[jj * ks + c] : 0.0;
  }
}

// DIRECT: the forward pass reads the columns of the row-major blocks themselves (8-byte loads, coalesced across
// the lanes) instead of the transposed copies - for small blocks the few extra load instructions cost less than
// the transpose launch (K <= 8: used by the acrobot / spinner / hopper models)
template <int K, bool DIRECT>
__global__ void __launch_bounds__(256)
penta_apply_kernel(int n, int k, const double* __restrict__ Ust, const double* __restrict__ Hst,
                   const double* __restrict__ Est, const double* __restrict__ Dst, const double* __restrict__ Tst,
                   const double* __restrict__ rhs, double rhs_sign, int nrhs, double* __restrict__ x, int m_split,
                   size_t cstride, RhsSource R) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // Two-sided factors: TWO wavefronts per right-hand side, one per chain (the halves of the twisted
  // factorisation are independent up to the join rows), which hand over through LDS: the bottom one its
  // pending pushes into the join rows, the top one x_m, x_{m+1} back.  2 (n/2) + 4 block rows on the
  // dependent chain instead of 2 n.  One-sided factors: one wavefront per right-hand side as before.
  const bool two = m_split > 0;
  const int wpc = two ? 2 : 1;                        // wavefronts per column
  const int half = two ? (wave & 1) : 0, slot = wave / wpc;
  const int j = blockIdx.x * ((blockDim.x >> 6) / wpc) + slot;
  constexpr int XCH = 4 * 64 + 2;                     // per column: q1, q2, x_m, x_{m+1} per lane; two flags
  double* colbase = lds + (size_t)slot * ((size_t)n * K + XCH);
  double* rtw = colbase;                              // rt_i[c] of this right-hand side, all rows (original index)
  double* exq = colbase + (size_t)n * K;              // [2][64] pending pushes of the bottom chain, then [2][64] x_m, x_{m+1}
  volatile int* xflag = reinterpret_cast<volatile int*>(exq + 4 * 64);   // [0] bottom forward done, [1] join rows solved
  if (two && half == 0 && lane < 2) xflag[lane] = 0;
  __syncthreads();                                    // (the only barrier: flags initialised)
  if (j >= nrhs) return;
  constexpr int ks = ldl_ks(K), KS2 = K * ks, KP = (K + 1) / 2;
  const size_t nk = cstride;  // distance between right-hand sides (>= n * k: the caller may solve a sub-system)
  const int c = (lane < K) ? lane : K - 1;  // lanes >= K shadow lane K-1 (never stored)
  const bool live = lane < K;
  // Two-sided factors (m_split > 0, see penta_ldl_kernel): block rows 0 .. m+1 belong to the
  // top-down recursion (rows m, m+1 are the join), rows n-1 .. m+2 to the mirrored bottom-up one,
  // whose local row il is original row n-1-il.  One wavefront walks both chains one after the
  // other; the bottom chain's pending pushes enter the join rows, x_m and x_{m+1} start its
  // back substitution.
  const int nT = two ? m_split + 2 : n, nB = two ? n - m_split - 2 : 0;

  // ---- forward substitution over local rows [first, last) of one chain.  The factor blocks
  // (transposed copies: 16-byte loads down the lane's column) and the right-hand side of the NEXT
  // row are loaded while this row's chain of v_readlane / FMA runs; two register sets alternate.
  const double* UT = Tst;
  const double* HT = Tst + (size_t)n * KS2;
  const double* ET = Tst + (size_t)2 * n * KS2;
  struct Blk { double2 u[KP], h[KP], e[KP]; double b; };
  // component c of block row i of this column's right-hand side
  const double* jslab = R.slab ? at_set(R.slab, R.alt) : nullptr;
  int jt = 0, jdof = 0;
  if (jslab && j > 0) { jt = (j - 1) / R.nu; jdof = R.dofs[(j - 1) - jt * R.nu]; }
  auto rhs_at = [&](int i) -> double {
    if (!jslab) return rhs[(size_t)j * nk + (size_t)i * k + c];
    const int gi = i + R.r0;
    if (j == 0) return R.g[(size_t)gi * k + c];
    const int which = gi - jt + 1;   // 0: q_{t-1}, 1: q_t, 2: q_{t+1}
    if (which < 0 || which > 2 || (which == 0 && jt < 2) || (which == 1 && jt < 1)) return 0.0;   // (q_0 is not a variable)
    return jslab[(size_t)jt * R.slab_stride + (size_t)which * R.nv * k + c * R.nv + jdof];
  };
  auto fload = [&](int side, int il, Blk& d) __attribute__((always_inline)) {
    const int i = side ? n - 1 - il : il;
    if (DIRECT) {
      const double* pu = Ust + (size_t)i * KS2 + c;   // element [jj][c] of the row-major block: jj * ks + c
      const double* ph = Hst + (size_t)i * KS2 + c;
      const double* pe = Est + (size_t)i * KS2 + c;
#pragma unroll
      for (int m = 0; m < KP; ++m) {
        const bool two_ = 2 * m + 1 < K;
        d.u[m] = make_double2(pu[(2 * m) * ks], two_ ? pu[(2 * m + 1) * ks] : 0.0);
        d.h[m] = make_double2(ph[(2 * m) * ks], two_ ? ph[(2 * m + 1) * ks] : 0.0);
        d.e[m] = make_double2(pe[(2 * m) * ks], two_ ? pe[(2 * m + 1) * ks] : 0.0);
      }
    } else {
      const double2* pu = reinterpret_cast<const double2*>(UT + (size_t)i * KS2 + c * ks);
      const double2* ph = reinterpret_cast<const double2*>(HT + (size_t)i * KS2 + c * ks);
      const double2* pe = reinterpret_cast<const double2*>(ET + (size_t)i * KS2 + c * ks);
#pragma unroll
      for (int m = 0; m < KP; ++m) { d.u[m] = pu[m]; d.h[m] = ph[m]; d.e[m] = pe[m]; }
    }
    d.b = (c < k) ? rhs_at(i) : 0.0;
  };
  auto fstep = [&](int side, int il, int last, const Blk& cur, Blk& nxt, double& pend1, double& pend2)
      __attribute__((always_inline)) {
    const int i = side ? n - 1 - il : il;
    double v = rhs_sign * cur.b + pend1;
    fload(side, il + 1 < last ? il + 1 : il, nxt);
    double a1 = pend2, a2 = 0.0;
#pragma unroll
    for (int jj = 0; jj < K; ++jj) {
      const double t = rdlane(v, jj);       // rt_i[jj]: final once the steps before it are done
      const double ujj = (jj & 1) ? cur.u[jj / 2].y : cur.u[jj / 2].x;
      const double hjj = (jj & 1) ? cur.h[jj / 2].y : cur.h[jj / 2].x;
      const double ejj = (jj & 1) ? cur.e[jj / 2].y : cur.e[jj / 2].x;
      v = __builtin_fma(-ujj, t, v);        // L[c][jj] = (D^-1 U)[jj][c], zero for c <= jj
      a1 = __builtin_fma(-hjj, t, a1);      // (Ht_i^T Dn rt_i)[c]
      a2 = __builtin_fma(-ejj, t, a2);      // (Et_i^T Dn rt_i)[c]
    }
    if (live) rtw[i * K + lane] = v;
    pend1 = a1;  // pushed into the next row of the chain
    pend2 = a2;  // and the one after it
  };
  auto forward = [&](int side, int first, int last, double& pend1, double& pend2) {
    if (first >= last) return;
    Blk A, B;
    fload(side, first, A);
    for (int il = first; il < last; il += 2) {
      fstep(side, il, last, A, B, pend1, pend2);
      if (il + 1 < last) fstep(side, il + 1, last, B, A, pend1, pend2);
    }
  };
  auto wait_flag = [&](int f) {
    while (xflag[f] == 0) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  };
  auto post_flag = [&](int f) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) xflag[f] = 1;
  };
  double p1 = 0.0, p2 = 0.0;
  if (half == 0) {
    forward(0, 0, two ? m_split : n, p1, p2);
    if (two) {
      wait_flag(0);
      p1 += exq[64 + lane];  // the bottom chain's "row nB + 1" is original row m, its "row nB" is row m+1
      p2 += exq[lane];
      forward(0, m_split, nT, p1, p2);
    }
  } else {
    double q1 = 0.0, q2 = 0.0;
    forward(1, 0, nB, q1, q2);
    exq[lane] = q1; exq[64 + lane] = q2;
    post_flag(0);
  }

  // ---- back substitution over local rows first, first-1, .., 0 of one chain
  // v: pending right-hand side of the row being solved; p: what has been pushed so far into the
  // row after it (-(D^-1 Et_{i-1}) x_{i+1}, pushed one iteration earlier).  `given` leading rows
  // are already solved (the join rows, seen from the bottom chain): they only push.
  const int r = c;
  double xm = 0.0, xm1 = 0.0;  // x_m, x_{m+1} (join rows)
  auto backward = [&](int side, int first, int given) {
    auto o = [&](int il) { const int t = side ? n - 1 - il : il; return t < 0 ? 0 : (t > n - 1 ? n - 1 : t); };
    double v = given ? xm : Dst[(size_t)o(first) * K + r] * rtw[o(first) * K + r];
    double p = 0.0;
    struct Bk { double2 u[KP], h[KP], e[KP]; double d; };
    auto bload = [&](int il, Bk& t) __attribute__((always_inline)) {
      const double2* pu = reinterpret_cast<const double2*>(Ust + (size_t)o(il) * KS2 + r * ks);
      const double2* ph = reinterpret_cast<const double2*>(Hst + (size_t)o(il > 0 ? il - 1 : 0) * KS2 + r * ks);
      const double2* pe = reinterpret_cast<const double2*>(Est + (size_t)o(il > 1 ? il - 2 : 0) * KS2 + r * ks);
#pragma unroll
      for (int m = KP - 1; m >= 0; --m) { t.u[m] = pu[m]; t.h[m] = ph[m]; t.e[m] = pe[m]; }
      t.d = Dst[(size_t)o(il > 0 ? il - 1 : 0) * K + r];
    };
    auto bstep = [&](int il, const Bk& cur, Bk& nxt) __attribute__((always_inline)) {
      const bool solved = il > first - given;
      bload(il > 0 ? il - 1 : 0, nxt);
      // rows -1, -2 do not exist; the join rows are not coupled THROUGH THIS CHAIN to each other
      const double hs = (il > 0 && !(given && il == first)) ? -1.0 : 0.0, es = (il > 1) ? -1.0 : 0.0;
      const double us = solved ? 0.0 : 1.0;
      const double next_rt = (il > 0) ? cur.d * rtw[o(il - 1) * K + r] : 0.0;
      double pn = 0.0;
#pragma unroll
      for (int jj = K - 1; jj >= 0; --jj) {
        const double xj = rdlane(v, jj);
        const double ujj = (jj & 1) ? cur.u[jj / 2].y : cur.u[jj / 2].x;
        const double hjj = (jj & 1) ? cur.h[jj / 2].y : cur.h[jj / 2].x;
        const double ejj = (jj & 1) ? cur.e[jj / 2].y : cur.e[jj / 2].x;
        v = __builtin_fma(-(ujj * us), xj, v); // strictly upper: rows >= jj keep their value
        p = __builtin_fma(hjj * hs, xj, p);    // row il-1: -(D^-1 Ht_{il-1}) x_il
        pn = __builtin_fma(ejj * es, xj, pn);  // row il-2: -(D^-1 Et_{il-2}) x_il
      }
      if (!solved) {
        if (live && lane < k) x[(size_t)j * nk + (size_t)o(il) * k + lane] = v;
        if (two && !side && il == m_split + 1) xm1 = v;
        if (two && !side && il == m_split) {   // both join rows solved: the bottom chain's back substitution can start
          xm = v;
          exq[128 + lane] = xm; exq[192 + lane] = xm1;
          post_flag(1);
        }
      }
      const bool next_given = il - 1 > first - given;  // (the second join row: what was pushed into it is not needed)
      v = next_given ? xm1 : next_rt + p;
      p = pn;
    };
    Bk A, B;
    bload(first, A);
    for (int il = first; il >= 0; il -= 2) {
      bstep(il, A, B);
      if (il >= 1) bstep(il - 1, B, A);
    }
  };
  if (half == 0) {
    if (jslab && lane < k)   // the block rows above the sub-system: x = rhs = 0 (g_0 = 0, J has no column of q_0)
      for (int i0 = 0; i0 < R.r0; ++i0) x[(size_t)j * nk + (size_t)(i0 - R.r0) * k + lane] = 0.0;
    backward(0, nT - 1, 0);
  } else {
    wait_flag(1);
    xm = exq[128 + lane]; xm1 = exq[192 + lane];
    backward(1, nB + 1, 2);
  }
}

}  // namespace idto_dev
