// kkt.h — the equality-constraint step of the resident trust-region loop as ONE banded solve
// (reference optimizer/trajectory_optimizer.cc:1292-1396 CalcEqualityConstraintJacobian / CalcLagrangeMultipliers and
// :2139-2149, the H^-1 (g + J^T lambda) of CalcDoglegPoint).
//
// The reference forms lambda = (J H^-1 J^T)^-1 (h - J H^-1 g) with a dense factorisation of the n_eq x n_eq Schur
// complement and then w = H^-1 (g + J^T lambda).  constraints.h does the same on the device: H^-1 [g | J^T]
// (factorisation + n_eq + 1 substitutions), S = J Y, a dense LDL^T in one workgroup, y_g + Y lambda - five launches,
// 100 of hopper's 170 us per iteration.  The same two vectors are the solution of the KKT system
//
//     [ H  J^T ] [  w       ]   [ g ]
//     [ J   0  ] [ -lambda  ] = [ h ]
//
// and J is as banded as H: row (t, dof) of J, the derivative of tau_t[dof], touches q_{t-1}, q_t, q_{t+1} only.  With
// the unknowns interleaved as z_t = [x_t ; mu_t], mu_t = -lambda_{t-1} (the multipliers of tau_{t-1}: ALL the
// variables tau_{t-1} depends on are in block rows <= t), the matrix is block penta-diagonal with blocks of nq + nu
// and the banded solvers of penta_ldl.h take it as it is:
//
//     M_{t,t}   = [ H_tt       J_{t-1,t}^T ]     M_{t,t-1} = [ H_{t,t-1}    0 ]     M_{t,t-2} = [ H_{t,t-2}    0 ]
//                 [ J_{t-1,t}  0           ]                 [ J_{t-1,t-1}  0 ]                 [ J_{t-1,t-2}  0 ]
//
// (J_{s,c} = d tau_s[dofs] / d q_c).  An unpivoted LDL^T meets nq positive pivots, then nu negative ones in every block
// (the multiplier rows' Schur complement is -J (..)^-1 J^T of the variables eliminated before them, which include q_t with
// its M / dt^2 - full row rank; the bottom-up chain of the two-sided elimination meets the blocks in the other order,
// the rows inside a block in the same one, and the same argument holds).  The solvers' pivot test covers the nq rows
// of H in every block (penta_ldl.h ldl_pivot_bad); the multiplier pivots are judged here.  Block row 0 is decoupled as in H (q_0 is no variable: its columns of
// J are zero, TO.cc:1316-1322), mu_0 is a dummy with an identity block.
//
// What this buys: no dense Schur complement (no n_eq <= 128 limit of the single-workgroup factorisation, no blocked
// dense LDL^T above it), no n_eq + 1 substitutions; and the conditioning is that of the KKT matrix, not of
// S = J H^-1 J^T.  Redundant constraints show as a multiplier pivot that cancels to nothing: kkt_extract_kernel raises
// TRF_SINGULAR_S from the range of the multiplier pivots, as constraint_lambda_kernel does from S's.
#pragma once
#include <hip/hip_runtime.h>

#include "batch.h"
#include "constraints.h"
#include "trust_region.h"

namespace idto_dev {

struct KktBuildArgs {
  int N, nq, nv, nu;
  const double *HA, *HB, *HC, *g;     // the Gauss-Newton Hessian's lower bands (blocks nq x nq, column-major) and the gradient
  const double* slab; int slab_stride;   // fd_kernel's records of the iterate (AltSel: whichever set holds it)
  const int* dofs;                    // [nu] constrained (unactuated) degrees of freedom
  double *KA, *KB, *KC, *rhs;         // out: bands of the KKT matrix (blocks (nq + nu)^2, column-major), [g_t ; h_{t-1}]
  AltSel alt;
  size_t pstride, kstride;            // batch (grid.y = problem): arena strides in bytes of the context and of the KKT context
};

// grid: (N + 1 workgroups (block row t), problems), any block size
__global__ void kkt_build_kernel(KktBuildArgs A) {
  const int t = blockIdx.x, nq = A.nq, nv = A.nv, nu = A.nu, N = A.N, K = nq + nu, kk = K * K, qq = nq * nq;
  const size_t o = (size_t)blockIdx.y * A.pstride, ok = (size_t)blockIdx.y * A.kstride;
  A.HA = at_problem(A.HA, o); A.HB = at_problem(A.HB, o); A.HC = at_problem(A.HC, o); A.g = at_problem(A.g, o);
  A.KA = at_problem(A.KA, ok); A.KB = at_problem(A.KB, ok); A.KC = at_problem(A.KC, ok); A.rhs = at_problem(A.rhs, ok);
  const double* slab = at_problem(A.slab, o + (size_t)alt_offset(A.alt, o));
  // a group of 32 lanes per (band, column) pair, lane = row (K <= 32); wider blocks: the general loop below.
  // (One flat index per entry cost two divisions by run-time numbers and a third inside jac_entry: 6.6 of the kernel's
  // 11.6 us at allegro's 29 x 29.)  The entries of J come straight from the slab's records: row (t - 1, dof) of J against
  // block column s = t - 2 + band is block `band` of record t - 1 (dtau_{t-1}/dq_{t-2}, /dq_{t-1}, /dq_t), and the
  // transposed entries are non-zero in the diagonal block only (jac_entry's rules: q_0 is no variable).
  if (K <= 32 && (blockDim.x & 31) == 0) {
    const int lane = threadIdx.x & 31, grp = threadIdx.x >> 5, ngrp = blockDim.x >> 5;
    const double* rec = (t >= 1) ? slab + (size_t)(t - 1) * A.slab_stride : slab;
    for (int bc = grp; bc < 3 * K; bc += ngrp) {
      const int band = bc / K, c = bc - band * K, s = t - 2 + band, r = lane;
      if (r >= K) continue;
      double v = 0.0;
      if (s >= 0) {
        if (r < nq && c < nq) {
          const double* Hb = (band == 0) ? A.HA : (band == 1) ? A.HB : A.HC;
          v = Hb[(size_t)t * qq + c * nq + r];
        } else if (r >= nq && c < nq) {   // mu_t's row: J_{t-1, s}[dof, c]
          const bool zero = t < 1 || (band == 0 && t - 1 < 2) || (band == 1 && t - 1 < 1);
          if (!zero) v = rec[(size_t)band * nv * nq + c * nv + A.dofs[r - nq]];
        } else if (r < nq) {              // mu_s's column: J_{s-1, t}^T, the diagonal block's only
          if (band == 2 && t >= 1) v = rec[(size_t)2 * nv * nq + r * nv + A.dofs[c - nq]];
        } else if (t == 0 && band == 2 && r == c) {
          v = 1.0;                        // the dummy mu_0
        }
      }
      double* Kb = (band == 0) ? A.KA : (band == 1) ? A.KB : A.KC;
      Kb[(size_t)t * kk + c * K + r] = v;
    }
  } else
  for (int idx = threadIdx.x; idx < 3 * kk; idx += blockDim.x) {
    const int band = idx / kk, e = idx - band * kk, c = e / K, r = e - c * K;   // band 0: M_{t,t-2}, 1: M_{t,t-1}, 2: M_{t,t}
    const int s = t - 2 + band;                                                // block column
    double v = 0.0;
    if (s >= 0) {
      if (r < nq && c < nq) {
        const double* Hb = (band == 0) ? A.HA : (band == 1) ? A.HB : A.HC;
        v = Hb[(size_t)t * qq + c * nq + r];
      } else if (r >= nq && c < nq) {   // mu_t's row: J_{t-1, s}
        if (t >= 1) v = jac_entry(slab, A.slab_stride, nq, nv, t - 1, A.dofs[r - nq], N, s * nq + c);
      } else if (r < nq) {              // mu_s's column: J_{s-1, t}^T (zero outside the diagonal block)
        if (s >= 1) v = jac_entry(slab, A.slab_stride, nq, nv, s - 1, A.dofs[c - nq], N, t * nq + r);
      } else if (t == 0 && band == 2 && r == c) {
        v = 1.0;                        // the dummy mu_0
      }
    }
    double* Kb = (band == 0) ? A.KA : (band == 1) ? A.KB : A.KC;
    Kb[(size_t)t * kk + e] = v;
  }
  for (int r = threadIdx.x; r < K; r += blockDim.x) {
    double v = 0.0;
    if (r < nq) v = A.g[(size_t)t * nq + r];
    else if (t >= 1) v = slab[(size_t)(t - 1) * A.slab_stride + 3 * nv * nq + A.dofs[r - nq]];   // h = tau_{t-1}[dof]
    A.rhs[(size_t)t * K + r] = v;
  }
}

struct KktExtractArgs {
  int N, nq, nv, nu;
  const double* z;                    // the solver's x for the right-hand side -[g ; h]: z_t = [-w_t ; lambda_{t-1}]
  const double* slab; int slab_stride;
  const int* dofs;
  double *w, *jtl, *lambda;           // out: H^-1 (g + J^T lambda) [(N+1) nq], J^T lambda [(N+1) nq], lambda [N nu]
  const double* Dinv; int dstride, first_row;   // the factorisation's 1 / d: solver row i (= block row i + first_row), lane r at Dinv[i dstride + r]
  double* state;                      // the loop's state (TRS_FLAGS)
  AltSel alt;
  size_t pstride, kstride;            // batch (grid.y = problem): arena strides in bytes of the context and of the KKT context
};

// grid: (N + 1 workgroups of 64 threads (block row t), problems); workgroup 0 also looks at the multiplier pivots
__global__ void __launch_bounds__(64) kkt_extract_kernel(KktExtractArgs A) {
  const int t = blockIdx.x, nq = A.nq, nv = A.nv, nu = A.nu, N = A.N, K = nq + nu, lane = threadIdx.x;
  const size_t o = (size_t)blockIdx.y * A.pstride, ok = (size_t)blockIdx.y * A.kstride;
  A.z = at_problem(A.z, ok); A.Dinv = at_problem(A.Dinv, ok);
  A.w = at_problem(A.w, o); A.jtl = at_problem(A.jtl, o); A.lambda = at_problem(A.lambda, o); A.state = at_problem(A.state, o);
  const double* slab = at_problem(A.slab, o + (size_t)alt_offset(A.alt, o));
  for (int r = lane; r < nq; r += 64) {
    const int i = t * nq + r;
    A.w[i] = -A.z[(size_t)t * K + r];
    double jt = 0.0;   // (constraint_step_kernel's sum: ascending time step, then dof)
    for (int s = (t >= 1 ? t - 1 : 0); s <= t + 1 && s < N; ++s)
      for (int j = 0; j < nu; ++j) jt += jac_entry(slab, A.slab_stride, nq, nv, s, A.dofs[j], N, i) * A.z[(size_t)(s + 1) * K + nq + j];
    A.jtl[i] = jt;
  }
  if (t >= 1)
    for (int j = lane; j < nu; j += 64) A.lambda[(size_t)(t - 1) * nu + j] = A.z[(size_t)t * K + nq + j];
  if (t == 0) {
    // the multiplier pivots (1 / d is what the factorisation keeps): negative, finite, and min |d| / max |d| > 1e-13 -
    // anything else = redundant constraints, the host's pivoted factorisation takes over (constraint_lambda_kernel's
    // criterion on the pivots of S)
    double imn = __builtin_inf(), imx = 0.0;
    bool finite = true;
    for (int idx = lane; idx < N * nu; idx += 64) {
      const int bt = 1 + idx / nu, j = idx - (bt - 1) * nu;
      const double iv = -A.Dinv[(size_t)(bt - A.first_row) * A.dstride + nq + j];
      finite = finite && __builtin_isfinite(iv) && iv > 0.0;
      imn = __builtin_fmin(imn, iv); imx = __builtin_fmax(imx, iv);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      imn = __builtin_fmin(imn, __shfl_xor(imn, off)); imx = __builtin_fmax(imx, __shfl_xor(imx, off));
    }
    const bool bad = __builtin_amdgcn_ballot_w64(!finite) != 0ull || !(imn > 1e-13 * imx);   // (|d|: max = 1 / imn, min = 1 / imx)
    if (lane == 0 && bad) A.state[TRS_FLAGS] = (double)((int)A.state[TRS_FLAGS] | TRF_SINGULAR_S);
  }
}

}  // namespace idto_dev
