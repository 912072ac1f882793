// constraints.h — device side of the equality-constraint step of the trust-region iteration
// (reference optimizer/trajectory_optimizer.cc:1292-1396 CalcEqualityConstraintJacobian /
// CalcLagrangeMultipliers and the H^-1 (g + J^T lambda) of CalcDoglegPoint :2139-2149).
//
// The constraint h(q) = [tau_t[dof] : t < N, dof unactuated] has the Jacobian rows
//   J[(t, dof), :] = [ dtau_t/dq_{t-1} (t > 1) | dtau_t/dq_t (t > 0) | dtau_t/dq_{t+1} ](dof, :)
// which are rows of the slab records fd_kernel wrote, so nothing about J ever crosses PCIe:
//   constraint_rhs_kernel    columns [g | J^T] for the multi-right-hand-side solve,
//   constraint_schur_kernel  S = J Y_J (n_eq x n_eq) and J y_g from Y = H^-1 [g | J^T],
//   constraint_step_kernel   y_g + Y_J lambda  (= H^-1 (g + J^T lambda))  and  J^T lambda.
#pragma once
#include <hip/hip_runtime.h>

// slab record k: [dtau_k/dq_{k-1} | dtau_k/dq_k | dtau_k/dq_{k+1} | tau_k], blocks nv x nq stored
// column by column (index i * nv + r)
__device__ __forceinline__ double jac_entry(const double* __restrict__ slab, int slab_stride, int nq, int nv, int t,
                                            int dof, int N, int col) {
  // J[(t, dof), col] for a global column index col in [0, (N+1) nq)
  const int tc = col / nq, i = col - tc * nq;
  const int which = tc - t + 1;  // 0: q_{t-1}, 1: q_t, 2: q_{t+1}
  if (which < 0 || which > 2) return 0.0;
  if ((which == 0 && t < 2) || (which == 1 && t < 1)) return 0.0;  // q_0 is not a variable of tau_t's rows (:1316-1322)
  return slab[(size_t)t * slab_stride + (size_t)which * nv * nq + i * nv + dof];
}

// grid: n_eq + 1 blocks; block 0 copies g, block 1 + r writes row r of J as a column
__global__ void constraint_rhs_kernel(const double* __restrict__ slab, int slab_stride, const double* __restrict__ g,
                                      const int* __restrict__ dofs, int nu, int N, int nq, int nv,
                                      double* __restrict__ rhs) {
  const int n = (N + 1) * nq, b = blockIdx.x;
  double* out = rhs + (size_t)b * n;
  if (b == 0) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) out[i] = g[i];
    return;
  }
  const int r = b - 1, t = r / nu, dof = dofs[r - t * nu];
  for (int i = threadIdx.x; i < n; i += blockDim.x) out[i] = jac_entry(slab, slab_stride, nq, nv, t, dof, N, i);
}

// grid: n_eq blocks (row r of J staged in LDS); thread <-> column of Y.  out_S is column-major
// n_eq x n_eq (S[r + c * n_eq] = J_r . Y_{1+c}); out_Jy[r] = J_r . y_g.  Products are summed in
// ascending column order of the three blocks around time step t.
__global__ void constraint_schur_kernel(const double* __restrict__ slab, int slab_stride, const int* __restrict__ dofs,
                                        int nu, int N, int nq, int nv, const double* __restrict__ Y, int neq,
                                        double* __restrict__ out_S, double* __restrict__ out_Jy) {
  extern __shared__ double jr[];  // [3 nq]
  const int n = (N + 1) * nq, r = blockIdx.x, t = r / nu, dof = dofs[r - t * nu];
  const int c0 = (t >= 1 ? t - 1 : 0) * nq, len = (t + 2) * nq - c0;
  for (int i = threadIdx.x; i < len; i += blockDim.x) jr[i] = jac_entry(slab, slab_stride, nq, nv, t, dof, N, c0 + i);
  __syncthreads();
  for (int c = threadIdx.x; c <= neq; c += blockDim.x) {
    const double* y = Y + (size_t)c * n + c0;
    double acc = 0.0;
    for (int i = 0; i < len; ++i) acc += jr[i] * y[i];
    if (c == 0) out_Jy[r] = acc;
    else out_S[(size_t)(c - 1) * neq + r] = acc;
  }
}

// workgroup <-> 64 consecutive variables i, STEP_WAVES wavefronts splitting the sum over r (the sum
// is a chain of dependent loads per thread: 16 short chains instead of one of n_eq terms):
// out_step[i] = y_g[i] + sum_r Y_{1+r}[i] lambda[r] (partial sums added in wavefront order);
// out_jtl[i] = sum_r J[r, i] lambda[r] (ascending r; only the rows of the three time steps around
// i's own are non-zero)
constexpr int STEP_WAVES = 16;
__global__ void __launch_bounds__(64 * STEP_WAVES)
constraint_step_kernel(const double* __restrict__ slab, int slab_stride, const int* __restrict__ dofs,
                       int nu, int N, int nq, int nv, const double* __restrict__ Y, int neq,
                       const double* __restrict__ lambda, double* __restrict__ out_step,
                       double* __restrict__ out_jtl) {
  extern __shared__ double lam[];  // [neq] + [STEP_WAVES][64] partial sums
  double* part = lam + neq;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int r = threadIdx.x; r < neq; r += blockDim.x) lam[r] = lambda[r];
  __syncthreads();
  const int n = (N + 1) * nq, i = blockIdx.x * 64 + lane;
  const int per = (neq + STEP_WAVES - 1) / STEP_WAVES, r0 = w * per, r1 = (r0 + per < neq) ? r0 + per : neq;
  double acc = 0.0;
  if (i < n)
    for (int r = r0; r < r1; ++r) acc += Y[(size_t)(1 + r) * n + i] * lam[r];
  part[w * 64 + lane] = acc;
  __syncthreads();
  if (i >= n) return;
  if (w == 0) {
    double tot = Y[i];
    for (int ww = 0; ww < STEP_WAVES; ++ww) tot += part[ww * 64 + lane];
    out_step[i] = tot;
  } else if (w == 1) {
    const int ti = i / nq;
    double jt = 0.0;
    for (int t = (ti >= 1 ? ti - 1 : 0); t <= ti + 1 && t < N; ++t)
      for (int j = 0; j < nu; ++j) jt += jac_entry(slab, slab_stride, nq, nv, t, dofs[j], N, i) * lam[t * nu + j];
    out_jtl[i] = jt;
  }
}
