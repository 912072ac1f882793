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

#include "batch.h"
#include "trust_region.h"

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
                                      double* __restrict__ rhs, double* __restrict__ x, idto_dev::AltSel alt) {
  slab = idto_dev::at_set(slab, alt);   // (idto_hip_tr_solve: the iterate's set of partials)
  const int n = (N + 1) * nq, b = blockIdx.x;
  double* out = rhs + (size_t)b * n;
  double* x0 = x + (size_t)b * n;       // block row 0 of the solution = of the right-hand side (C_0 = I, decoupled):
  if (b == 0) {                         // written here, the solver starts at row 1
    for (int i = threadIdx.x; i < n; i += blockDim.x) { const double v = g[i]; out[i] = v; if (i < nq) x0[i] = v; }
    return;
  }
  const int r = b - 1, t = r / nu, dof = dofs[r - t * nu];
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const double v = jac_entry(slab, slab_stride, nq, nv, t, dof, N, i);
    out[i] = v;
    if (i < nq) x0[i] = v;
  }
}

// grid: n_eq blocks (row r of J staged in LDS); thread <-> column of Y.  out_S is column-major
// n_eq x n_eq (S[r + c * n_eq] = J_r . Y_{1+c}); out_Jy[r] = J_r . y_g.  Products are summed in
// ascending column order of the three blocks around time step t.
__global__ void constraint_schur_kernel(const double* __restrict__ slab, int slab_stride, const int* __restrict__ dofs,
                                        int nu, int N, int nq, int nv, const double* __restrict__ Y, int neq,
                                        double* __restrict__ out_S, double* __restrict__ out_Jy, idto_dev::AltSel alt) {
  slab = idto_dev::at_set(slab, alt);
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
                       double* __restrict__ out_jtl, idto_dev::AltSel alt) {
  slab = idto_dev::at_set(slab, alt);
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


// ---------------------------------------------------------------------------
// constraint_lambda_kernel: lambda = S^-1 (h - J y_g) (TO.cc:1371-1396) in ONE workgroup for small
// n_eq (S in LDS: n_eq <= CON_LAMBDA_MAX), so that the device-resident trust-region loop
// (idto_hip_tr_solve) needs neither the host's factorisation nor the ~2 launches per 32 columns of
// dense_ldl.h.  h = tau_t[dof] of the iterate comes straight from the slab.  Unpivoted LDL^T of the
// symmetric positive definite S = J H^-1 J^T, right-looking by panels of four columns, the lower
// triangle in 4 x 4 register tiles (one per thread, 528 of them): per panel the owner of the diagonal
// tile factorises it in registers, the tiles below it solve for their rows of L, everybody applies the
// rank-4 update (32 LDS reads, 64 multiply-adds) - two barriers per panel, and the forward substitution
// of the right-hand side rides along in the diagonal tiles' owners.  (An element-by-element update of an
// LDS-resident triangle took 254 us at n_eq = 120, one column per barrier 119 us.)  The backward
// substitution runs on one wavefront (lane = 2 rows) without barriers.
// A pivot that is not positive, or min / max pivot <= 1e-13 (redundant constraints: the host's
// pivoted LDL^T copes, this does not), raises TRF_SINGULAR_S in the loop state: the remaining
// iterations idle and the host takes over from the iterate.
constexpr int CON_LAMBDA_MAX = 128;
constexpr int CON_TILE = 4;                                    // register tile = panel width of the factorisation
constexpr int CON_TGRID = CON_LAMBDA_MAX / CON_TILE;           // 32 x 32 tiles, the lower triangle has 528
constexpr int CON_LAMBDA_THREADS = 576;                        // >= 528, nine wavefronts
constexpr int CON_PS = CON_TILE + 1;                           // row stride of the published panel: odd, or the tile rows of
                                                               // consecutive lanes (4 rows = 128 B apart) meet in 2 of the 64 banks
__host__ __device__ constexpr int con_lambda_lds_doubles(int n) {
  return n * (n | 1) + 2 * CON_LAMBDA_MAX * CON_PS + CON_TILE * CON_TILE + 3 * CON_TILE + CON_LAMBDA_MAX;
}
__global__ void __launch_bounds__(CON_LAMBDA_THREADS)
constraint_lambda_kernel(const double* __restrict__ S_g /* [neq*neq | Jy] */, int neq, const double* __restrict__ slab,
                         int slab_stride, int tau_off, const int* __restrict__ dofs, int nu, double* __restrict__ lambda,
                         double* __restrict__ state, idto_dev::AltSel alt, double* __restrict__ stamps) {
  extern __shared__ double lds[];
  slab = idto_dev::at_set(slab, alt);
  if (stamps && threadIdx.x == 0) stamps[0] = (double)clock64();
  constexpr int TB = CON_TILE;
  const int tid = threadIdx.x, n = neq, np = (n + TB - 1) / TB;
  const int ld = n | 1;                   // odd column stride of L: columns of a row spread over the banks
  double* L = lds;                        // [n][ld] column-major: unit lower L below the diagonal, D on it
  constexpr int PS = CON_PS;
  double* Wp = L + (size_t)n * ld;        // [CON_LAMBDA_MAX][PS] the panel's columns below the diagonal tile, W = L D ...
  double* Lp = Wp + CON_LAMBDA_MAX * PS;  // [CON_LAMBDA_MAX][PS] ... and L itself
  double* Ld = Lp + CON_LAMBDA_MAX * PS;  // [TB][TB] unit lower factor of the diagonal tile, [TB] 1 / d, [2][TB] y of the panel
  double* zb = Ld + TB * TB + 3 * TB;     // (by panel parity: the next diagonal tile is factorised while phase C still reads y)
                                          // [CON_LAMBDA_MAX] D^-1 L^-1 (h - J y_g)
  __shared__ double dmin_s, dmax_s;
  // thread <-> tile (ti, tj), ti >= tj, of the lower triangle, in registers for the whole factorisation;
  // the owners of the diagonal tiles also carry their four rows of the right-hand side h - J y_g
  int ti = 0, tj = 0;
  {
    int rem = tid;
    while (tj < CON_TGRID && rem >= CON_TGRID - tj) { rem -= CON_TGRID - tj; ++tj; }
    ti = tj + rem;
  }
  const int i0 = ti * TB, k0 = tj * TB;
  const bool have = tj < CON_TGRID && ti < np;   // (tiles beyond the matrix idle)
  double a[TB][TB], b[TB];
#pragma unroll
  for (int r = 0; r < TB; ++r) {
    const int i = i0 + r;
#pragma unroll
    for (int c = 0; c < TB; ++c) {
      const int k = k0 + c;
      a[r][c] = (have && i < n && k < n) ? S_g[(size_t)k * n + i] : ((i == k) ? 1.0 : 0.0);   // (identity padding)
    }
    b[r] = 0.0;
    if (have && ti == tj && i < n) {
      const int t = i / nu, j = i - t * nu;
      b[r] = slab[(size_t)t * slab_stride + tau_off + dofs[j]] - S_g[(size_t)n * n + i];   // h - J y_g
    }
  }
  if (tid == 0) { dmin_s = __builtin_inf(); dmax_s = 0.0; }
  __syncthreads();
  if (stamps && tid == 0) stamps[1] = (double)clock64();
  // right-looking LDL^T by panels of TB columns, two barriers per panel; the forward substitution rides along
  for (int p = 0; p < np; ++p) {
    if (stamps && tid == 0 && p < 40) stamps[8 + p] = (double)clock64();
    const int j0 = p * TB;
    if (have && ti == p && tj == p) {   // A: the diagonal tile in registers: a = l d l^T, y = l^-1 b
      double dmn = dmin_s, dmx = dmax_s;
#pragma unroll
      for (int c = 0; c < TB; ++c) {
        const double d = a[c][c];
        double inv = __builtin_amdgcn_rcp(d);   // 1 / d: hardware estimate + 2 Newton steps (an IEEE division is ~4x
        double e = __builtin_fma(-d, inv, 1.0);  // as long, and four of them are the serial part of every panel)
        inv = __builtin_fma(inv, e, inv);
        e = __builtin_fma(-d, inv, 1.0);
        inv = __builtin_fma(inv, e, inv);
        if (j0 + c < n) { dmn = __builtin_fmin(dmn, d); dmx = __builtin_fmax(dmx, d); L[(size_t)(j0 + c) * ld + j0 + c] = d; }
        Ld[TB * TB + c] = inv;
        Ld[TB * TB + TB + (p & 1) * TB + c] = b[c];    // y_c: every earlier column has been applied
        zb[j0 + c] = b[c] * inv;
        double w[TB];
#pragma unroll
        for (int r = c + 1; r < TB; ++r) w[r] = a[r][c];   // l d
#pragma unroll
        for (int r = c + 1; r < TB; ++r) {
          const double l = w[r] * inv;
          a[r][c] = l;
          Ld[r * TB + c] = l;
          if (j0 + r < n) L[(size_t)(j0 + c) * ld + j0 + r] = l;
          b[r] = __builtin_fma(-l, b[c], b[r]);   // (explicit fma throughout: the library is built with -ffp-contract=off)
#pragma unroll
          for (int c2 = c + 1; c2 <= r; ++c2) a[r][c2] = __builtin_fma(-l, w[c2], a[r][c2]);
        }
      }
      dmin_s = dmn; dmax_s = dmx;
    }
    __syncthreads();
    if (stamps && tid == 0 && p < 40) stamps[64 + 2 * p] = (double)clock64();
    if (have && tj == p && ti > p) {    // B: the tiles below it: W = A l^-T (= L D), L = W D^-1
      double l[TB][TB], inv[TB];
#pragma unroll
      for (int c = 0; c < TB; ++c) {
        inv[c] = Ld[TB * TB + c];
#pragma unroll
        for (int c2 = 0; c2 < c; ++c2) l[c][c2] = Ld[c * TB + c2];
      }
#pragma unroll
      for (int r = 0; r < TB; ++r) {
#pragma unroll
        for (int c = 0; c < TB; ++c) {
          double w = a[r][c];
#pragma unroll
          for (int c2 = 0; c2 < c; ++c2) w = __builtin_fma(-a[r][c2], l[c][c2], w);   // (a[r][c2] already holds W(r, c2))
          a[r][c] = w;
        }
#pragma unroll
        for (int c = 0; c < TB; ++c) {
          const double lv = a[r][c] * inv[c];
          Wp[(i0 + r) * PS + c] = a[r][c];
          Lp[(i0 + r) * PS + c] = lv;
          if (i0 + r < n && j0 + c < n) L[(size_t)(j0 + c) * ld + i0 + r] = lv;
        }
      }
    }
    __syncthreads();
    if (stamps && tid == 0 && p < 40) stamps[65 + 2 * p] = (double)clock64();
    if (have && tj > p) {               // C: rank-TB update of the trailing tiles (and of the right-hand side)
      double lr[TB][TB], wk[TB][TB];
#pragma unroll
      for (int r = 0; r < TB; ++r)
#pragma unroll
        for (int m = 0; m < TB; ++m) { lr[r][m] = Lp[(i0 + r) * PS + m]; wk[r][m] = Wp[(k0 + r) * PS + m]; }
#pragma unroll
      for (int r = 0; r < TB; ++r)
#pragma unroll
        for (int c = 0; c < TB; ++c) {
          double acc = a[r][c];
#pragma unroll
          for (int m = 0; m < TB; ++m) acc = __builtin_fma(-lr[r][m], wk[c][m], acc);
          a[r][c] = acc;
        }
      if (ti == tj) {
#pragma unroll
        for (int r = 0; r < TB; ++r)
#pragma unroll
          for (int m = 0; m < TB; ++m) b[r] = __builtin_fma(-lr[r][m], Ld[TB * TB + TB + (p & 1) * TB + m], b[r]);
      }
    }
  }
  __syncthreads();
  if (stamps && tid == 0) stamps[2] = (double)clock64();
  // L^T x = z on one wavefront: lane owns rows lane and lane + 64
  if (tid < 64) {
    const int r0 = tid, r1 = tid + 64;
    double y0 = (r0 < n) ? zb[r0] : 0.0, y1 = (r1 < n) ? zb[r1] : 0.0;
    for (int j = n - 1; j >= 0; --j) {
      const double xj = (j < 64) ? __shfl(y0, j) : __shfl(y1, j - 64);   // final once every row below has been subtracted
      if (r0 < j) y0 = __builtin_fma(-L[(size_t)r0 * ld + j], xj, y0);   // L(j, r0)
      if (r1 < j && r1 < n) y1 = __builtin_fma(-L[(size_t)r1 * ld + j], xj, y1);
    }
    if (r0 < n) lambda[r0] = y0;
    if (r1 < n) lambda[r1] = y1;
    if (tid == 0 && (!(dmin_s > 1e-13 * dmax_s) || !__builtin_isfinite(dmax_s)))
      state[idto_dev::TRS_FLAGS] = (double)((int)state[idto_dev::TRS_FLAGS] | idto_dev::TRF_SINGULAR_S);
    if (stamps && tid == 0) stamps[3] = (double)clock64();
  }
}

// Larger constraint sets in the device-resident loop go through the blocked factorisation of dense_ldl.h:
// constraint_h_kernel forms its right-hand-side input [min pivot, max pivot | h] from the iterate's slab,
// constraint_flag_kernel turns the pivot range it leaves behind into TRF_SINGULAR_S.
__global__ void constraint_h_kernel(const double* __restrict__ slab, int slab_stride, int tau_off,
                                    const int* __restrict__ dofs, int nu, int neq, double* __restrict__ stat_h,
                                    idto_dev::AltSel alt) {
  slab = idto_dev::at_set(slab, alt);
  if (blockIdx.x == 0 && threadIdx.x == 0) { stat_h[0] = __builtin_inf(); stat_h[1] = 0.0; }
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < neq; r += gridDim.x * blockDim.x) {
    const int t = r / nu, j = r - t * nu;
    stat_h[2 + r] = slab[(size_t)t * slab_stride + tau_off + dofs[j]];
  }
}
__global__ void constraint_flag_kernel(const double* __restrict__ stat, double* __restrict__ state) {
  const double dmin = stat[0], dmax = stat[1];
  if (!(dmin > 1e-13 * dmax) || !__builtin_isfinite(dmax))
    state[idto_dev::TRS_FLAGS] = (double)((int)state[idto_dev::TRS_FLAGS] | idto_dev::TRF_SINGULAR_S);
}
