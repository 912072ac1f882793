// constraints.h — device side of the equality-constraint step of the trust-region iteration
// (reference optimizer/trajectory_optimizer.cc:1292-1396 CalcEqualityConstraintJacobian /
// CalcLagrangeMultipliers and the H^-1 (g + J^T lambda) of CalcDoglegPoint :2139-2149).
//
// The constraint h(q) = [tau_t[dof] : t < N, dof unactuated] has the Jacobian rows
//   J[(t, dof), :] = [ dtau_t/dq_{t-1} (t > 1) | dtau_t/dq_t (t > 0) | dtau_t/dq_{t+1} ](dof, :)
// which are rows of the slab records fd_kernel wrote, so nothing about J ever crosses PCIe:
//   (penta_apply.h RhsSource) the columns [g | J^T] of the multi-right-hand-side solve, read in place,
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
    int i = 0;
    for (; i + 8 <= len; i += 8) {   // (eight loads in flight: a rolled loop waits for every one of them; same order of adds)
      double y8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) y8[u] = y[i + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += jr[i + u] * y8[u];
    }
    for (; i < len; ++i) acc += jr[i] * y[i];
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
  if (i < n) {
    int r = r0;
    for (; r + 8 <= r1; r += 8) {    // (as above)
      double y8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) y8[u] = Y[(size_t)(1 + r + u) * n + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += y8[u] * lam[r + u];
    }
    for (; r < r1; ++r) acc += Y[(size_t)(1 + r) * n + i] * lam[r];
  }
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
// n_eq (the factor of S in LDS: n_eq <= CON_LAMBDA_MAX), so that the device-resident trust-region loop
// (idto_hip_tr_solve) needs neither the host's factorisation nor the ~2 launches per 32 columns of
// dense_ldl.h.  h = tau_t[dof] of the iterate comes straight from the slab.  Unpivoted LDL^T of the
// symmetric positive definite S = J H^-1 J^T, LEFT-looking by panels of four columns and WITHOUT
// barriers: eight wavefronts own the panels cyclically (lane = row, two rows per lane), a panel
// collects the updates of the panels before it as their flags appear in LDS, is factorised and
// published; a ninth wavefront follows with the right-hand side (forward substitution panel by
// panel), then substitutes backwards in blocks of four.  The chain per panel is "update from the
// previous panel + 4 x 4 factorisation + scaling" on one wavefront (~1.4k cycles) while the other
// seven apply older updates.  History at n_eq = 120: element-wise updates of an LDS-resident triangle
// 254 us; register tiles, one column per barrier 119 us; panels with two barriers 84 us (each phase
// transition cost ~2k cycles of a lone wavefront's latency); this version: see DESIGN.md 13.
// A pivot range min / max <= 1e-13 (redundant constraints: the host's pivoted LDL^T copes, this does
// not) raises TRF_SINGULAR_S in the loop state: the remaining iterations idle, the host takes over.
constexpr int CON_LAMBDA_MAX = 128;
constexpr int CON_TILE = 4;                                    // panel width
constexpr int CON_FWAVES = 8;                                  // wavefronts that factorise (+ 1 for the right-hand side)
constexpr int CON_LAMBDA_THREADS = 64 * (CON_FWAVES + 1);
constexpr int CON_MAXOWN = (CON_LAMBDA_MAX / CON_TILE + CON_FWAVES - 1) / CON_FWAVES;   // panels per wavefront
__host__ __device__ constexpr int con_lambda_ld(int n) { return (((n + CON_TILE - 1) / CON_TILE) * CON_TILE) | 1; }
__host__ __device__ constexpr int con_lambda_lds_doubles(int n) {
  return ((n + CON_TILE - 1) / CON_TILE) * CON_TILE * con_lambda_ld(n)      // L
         + (CON_FWAVES + 1) * CON_TILE * CON_TILE                           // per-wavefront exchange of a diagonal tile / 4 values
         + CON_LAMBDA_MAX                                                   // z
         + CON_LAMBDA_MAX / 2;                                              // panel flags (ints)
}
__global__ void __launch_bounds__(CON_LAMBDA_THREADS)
constraint_lambda_kernel(const double* __restrict__ S_g /* [neq*neq | Jy] */, int neq, const double* __restrict__ slab,
                         int slab_stride, int tau_off, const int* __restrict__ dofs, int nu, double* __restrict__ lambda,
                         double* __restrict__ state, idto_dev::AltSel alt, double* __restrict__ stamps) {
  extern __shared__ double lds[];
  slab = idto_dev::at_set(slab, alt);
  constexpr int TB = CON_TILE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = neq, np = (n + TB - 1) / TB, ld = con_lambda_ld(n);
  double* L = lds;                                   // [4 np][ld] column-major: unit lower L below the diagonal, D on it
  double* xch = L + TB * np * ld + wave * TB * TB;   // this wavefront's exchange area
  double* zb = L + TB * np * ld + (CON_FWAVES + 1) * TB * TB;   // [CON_LAMBDA_MAX] D^-1 L^-1 (h - J y_g), then lambda
  volatile int* flag = reinterpret_cast<volatile int*>(zb + CON_LAMBDA_MAX);   // [np] panel p is in L
  if (stamps && tid == 0) stamps[0] = (double)clock64();
  for (int p = tid; p < np; p += blockDim.x) flag[p] = 0;
  __syncthreads();
  auto wait_panel = [&](int q) {
    while (flag[q] == 0) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  };
  // the 4 x 4 block W(4p + c, 4q + m) = L D of panel q's columns at panel p's rows (wave-uniform)
  auto wblock = [&](int p, int q, double (&wb)[TB][TB]) {
#pragma unroll
    for (int m = 0; m < TB; ++m) {
      const double dm = L[(TB * q + m) * ld + TB * q + m];
#pragma unroll
      for (int c = 0; c < TB; ++c) wb[c][m] = L[(TB * q + m) * ld + TB * p + c] * dm;
    }
  };
  if (wave < CON_FWAVES) {
    // ---- factorisation: my panels are wave, wave + 8, ...; rows lane and lane + 64
    double a[CON_MAXOWN][2][TB];
#pragma unroll
    for (int k = 0; k < CON_MAXOWN; ++k) {
      const int p = wave + k * CON_FWAVES;
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        const int r = lane + 64 * sl;
#pragma unroll
        for (int c = 0; c < TB; ++c) {
          const int col = TB * p + c;
          a[k][sl][c] = (p < np && r < n && col < n && r >= TB * p) ? S_g[(size_t)col * n + r]
                                                                   : ((r == col) ? 1.0 : 0.0);   // (identity padding)
        }
      }
    }
#pragma unroll
    for (int k = 0; k < CON_MAXOWN; ++k) {
      const int p = wave + k * CON_FWAVES;
      if (p < np) {
        // updates from every earlier panel, as they are published
        for (int q = 0; q < p; ++q) {
          wait_panel(q);
          double wb[TB][TB];
          wblock(p, q, wb);
#pragma unroll
          for (int sl = 0; sl < 2; ++sl) {
            const int r = lane + 64 * sl;
            if (r >= TB * p && r < TB * np) {
              double lr[TB];
#pragma unroll
              for (int m = 0; m < TB; ++m) lr[m] = L[(TB * q + m) * ld + r];
#pragma unroll
              for (int c = 0; c < TB; ++c)
#pragma unroll
                for (int m = 0; m < TB; ++m) a[k][sl][c] = __builtin_fma(-lr[m], wb[c][m], a[k][sl][c]);
            }
          }
        }
        // the diagonal tile: its four rows live in four lanes - through LDS to everybody, factorised by everybody
        {
          const int r0 = TB * p, sl0 = r0 >> 6, l0 = r0 & 63;
          if (lane >= l0 && lane < l0 + TB) {
#pragma unroll
            for (int c = 0; c < TB; ++c) xch[(lane - l0) * TB + c] = (sl0 == 0) ? a[k][0][c] : a[k][1][c];
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          double t[TB][TB], inv[TB];
#pragma unroll
          for (int r = 0; r < TB; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c) t[r][c] = xch[r * TB + c];
#pragma unroll
          for (int c = 0; c < TB; ++c) {
            const double d = t[c][c];
            double iv = __builtin_amdgcn_rcp(d);   // 1 / d: hardware estimate + 2 Newton steps
            double e = __builtin_fma(-d, iv, 1.0);
            iv = __builtin_fma(iv, e, iv);
            e = __builtin_fma(-d, iv, 1.0);
            iv = __builtin_fma(iv, e, iv);
            inv[c] = iv;
            double w[TB];
#pragma unroll
            for (int r = c + 1; r < TB; ++r) w[r] = t[r][c];   // l d
#pragma unroll
            for (int r = c + 1; r < TB; ++r) {
              const double l = w[r] * iv;
              t[r][c] = l;
#pragma unroll
              for (int c2 = c + 1; c2 <= r; ++c2) t[r][c2] = __builtin_fma(-l, w[c2], t[r][c2]);
            }
          }
          // my rows of the panel: below the tile W = A l^-T, L = W D^-1; inside it l and d themselves
#pragma unroll
          for (int sl = 0; sl < 2; ++sl) {
            const int r = lane + 64 * sl;
            if (r >= r0 + TB && r < TB * np) {
#pragma unroll
              for (int c = 0; c < TB; ++c) {
                double w = a[k][sl][c];
#pragma unroll
                for (int c2 = 0; c2 < c; ++c2) w = __builtin_fma(-a[k][sl][c2], t[c][c2], w);   // (a[..][c2] already holds W)
                a[k][sl][c] = w;
              }
#pragma unroll
              for (int c = 0; c < TB; ++c) L[(r0 + c) * ld + r] = a[k][sl][c] * inv[c];
            } else if (r >= r0 && r < r0 + TB) {
#pragma unroll
              for (int c = 0; c < TB; ++c) {
#pragma unroll
                for (int rr = 0; rr < TB; ++rr)
                  if (r - r0 == rr && c <= rr) L[(r0 + c) * ld + r] = t[rr][c];   // d on the diagonal, l below it
              }
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          if (lane == 0) flag[p] = 1;
        }
      }
    }
    return;
  }
  // ---- the ninth wavefront: right-hand side.  rows lane and lane + 64 of h - J y_g
  double b[2];
#pragma unroll
  for (int sl = 0; sl < 2; ++sl) {
    const int r = lane + 64 * sl;
    b[sl] = 0.0;
    if (r < n) {
      const int t = r / nu, j = r - t * nu;
      b[sl] = slab[(size_t)t * slab_stride + tau_off + dofs[j]] - S_g[(size_t)n * n + r];
    }
  }
  double dmn = __builtin_inf(), dmx = 0.0;
  if (stamps && lane == 0) stamps[1] = (double)clock64();
  for (int q = 0; q < np; ++q) {   // forward: y_q = l^-1 b_q, z_q = y_q / d_q, b -= L(:, q) y_q
    wait_panel(q);
    const int r0 = TB * q, sl0 = r0 >> 6, l0 = r0 & 63;
    if (lane >= l0 && lane < l0 + TB) xch[lane - l0] = (sl0 == 0) ? b[0] : b[1];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    double y[TB];
#pragma unroll
    for (int c = 0; c < TB; ++c) {
      y[c] = xch[c];
#pragma unroll
      for (int c2 = 0; c2 < c; ++c2) y[c] = __builtin_fma(-L[(r0 + c2) * ld + r0 + c], y[c2], y[c]);
    }
#pragma unroll
    for (int c = 0; c < TB; ++c) {
      const double d = L[(r0 + c) * ld + r0 + c];
      if (r0 + c < n) { dmn = __builtin_fmin(dmn, d); dmx = __builtin_fmax(dmx, d); }
      if (lane == 0) zb[r0 + c] = y[c] / d;
    }
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
      const int r = lane + 64 * sl;
      if (r >= r0 + TB && r < TB * np) {
#pragma unroll
        for (int m = 0; m < TB; ++m) b[sl] = __builtin_fma(-L[(r0 + m) * ld + r], y[m], b[sl]);
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  if (stamps && lane == 0) stamps[2] = (double)clock64();
  // backward: L^T x = z in blocks of four from the bottom; x replaces z
  double x[2];
#pragma unroll
  for (int sl = 0; sl < 2; ++sl) { const int r = lane + 64 * sl; x[sl] = (r < TB * np) ? zb[r] : 0.0; }
  for (int q = np - 1; q >= 0; --q) {
    const int r0 = TB * q, sl0 = r0 >> 6, l0 = r0 & 63;
    if (lane >= l0 && lane < l0 + TB) xch[lane - l0] = (sl0 == 0) ? x[0] : x[1];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    double xs[TB];
#pragma unroll
    for (int c = TB - 1; c >= 0; --c) {   // unit upper l^T of the diagonal tile
      xs[c] = xch[c];
#pragma unroll
      for (int c2 = c + 1; c2 < TB; ++c2) xs[c] = __builtin_fma(-L[(r0 + c) * ld + r0 + c2], xs[c2], xs[c]);
    }
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
      const int r = lane + 64 * sl;
      if (r < r0) {
#pragma unroll
        for (int m = 0; m < TB; ++m) x[sl] = __builtin_fma(-L[r * ld + r0 + m], xs[m], x[sl]);   // L(r0 + m, r)
      } else if (r < r0 + TB) {
#pragma unroll
        for (int m = 0; m < TB; ++m)
          if (r - r0 == m) x[sl] = xs[m];
      }
    }
  }
#pragma unroll
  for (int sl = 0; sl < 2; ++sl) { const int r = lane + 64 * sl; if (r < n) lambda[r] = x[sl]; }
  if (lane == 0 && (!(dmn > 1e-13 * dmx) || !__builtin_isfinite(dmx)))
    state[idto_dev::TRS_FLAGS] = (double)((int)state[idto_dev::TRS_FLAGS] | idto_dev::TRF_SINGULAR_S);
  if (stamps && lane == 0) stamps[3] = (double)clock64();
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
