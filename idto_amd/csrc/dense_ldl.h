// dense_ldl.h — S lambda = r for the dense symmetric positive definite Schur complement
// S = J H^-1 J^T of the Lagrange multipliers (reference optimizer/trajectory_optimizer.cc:1395:
// Eigen ldlt() of an n_eq x n_eq matrix, n_eq = 40 .. 360 for the example models), on the device
// so that S (up to 1 MB) never crosses PCIe and the 8 MFLOP factorisation does not sit on one
// host core.
//
// Right-looking blocked LDL^T without pivoting, panels of NB = 32 columns, S column-major with
// only the lower triangle referenced:
//   dense_ldl_panel_kernel   (1 workgroup)  factorises the NB x NB diagonal block in LDS, solves
//                                           the panel rows below it (one thread per row), writes
//                                           L into S and W = L D into a workspace, tracks the
//                                           smallest / largest pivot;
//   dense_ldl_update_kernel  (many)         S[r][c] -= sum_k W[r][k] L[c][k] on 32 x 32 tiles of
//                                           the trailing lower triangle;
//   dense_ldl_solve_kernel   (1 workgroup)  L y = r, D z = y, L^T x = z, blocked the same way.
// No pivoting: S is positive definite whenever the constraint Jacobian has full row rank.  The
// pivot extremes are returned; the caller falls back to the host's pivoted LDL^T (which, like
// Eigen's, tolerates semi-definite S) when min pivot <= 1e-13 * max pivot or a pivot is not finite.
#pragma once
#include <hip/hip_runtime.h>

#include "penta_ldl.h"  // rdlane

namespace idto_dev {

constexpr int DENSE_NB = 32;

// stat[0] = min pivot so far, stat[1] = max pivot so far (initialised by the caller: +inf, 0).
// grid: 1 + ceil(rows below the block / 64) workgroups of one wavefront.  Every workgroup
// factorises the NB x NB diagonal block for itself in LDS (cheap, and it saves a launch);
// workgroup 0 writes the block's factor, the pivots and the statistics, workgroup b > 0 solves
// the 64 panel rows j0 + NB + 64 (b - 1) .. with one thread per row.
__global__ void __launch_bounds__(64)
dense_ldl_panel_kernel(double* __restrict__ S, int n, int j0, double* __restrict__ W, double* __restrict__ dvec,
                       double* __restrict__ stat) {
  constexpr int NB = DENSE_NB;
  __shared__ double A[NB][NB + 1];  // diagonal block, then its unit-lower factor
  __shared__ double dd[NB];
  const int tid = threadIdx.x, nt = 64;
  const int nb = (n - j0 < NB) ? n - j0 : NB;
  for (int idx = tid; idx < NB * NB; idx += nt) {
    const int r = idx % NB, c = idx / NB;
    A[r][c] = (r < nb && c < nb && r >= c) ? S[(size_t)(j0 + c) * n + j0 + r] : 0.0;
  }
  __syncthreads();
  // unpivoted LDL^T of the block in registers: lane r keeps row r; the pivot and the scaled pivot
  // column travel through v_readlane with constant lane numbers (fully unrolled: 496 updates).
  // LDS read-modify-write loops would be latency-bound (~250 cycles per element).
  {
    const int rr = tid & 31;
    double a[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) a[c] = A[rr][c];
#pragma unroll
    for (int p = 0; p < NB; ++p) {
      if (p < nb) {  // (wavefront-uniform)
        const double d = rdlane(a[p], p);
        const double l = a[p] * fast_rcp(d);  // l_rp on lane r > p
#pragma unroll
        for (int c = p + 1; c < NB; ++c) a[c] = __builtin_fma(-a[p], rdlane(l, c), a[c]);  // rows r >= c matter
        if (rr > p) a[p] = l;
      }
    }
    __syncthreads();
    if (tid < NB) {
#pragma unroll
      for (int c = 0; c < NB; ++c) A[rr][c] = a[c];
    }
  }
  __syncthreads();
  __shared__ double di[NB];
  if (tid < NB) {  // dd[r] = A[r][r] (static indexing above is not possible: read it back)
    dd[tid] = A[tid][tid];
    di[tid] = fast_rcp(A[tid][tid]);
  }
  __syncthreads();
  if (blockIdx.x == 0) {
    if (tid == 0) {
      double mn = stat[0], mx = stat[1];
      for (int p = 0; p < nb; ++p) {
        const double d = dd[p];
        if (!(d > mn)) mn = d;  // also catches NaN
        if (d > mx) mx = d;
      }
      stat[0] = mn; stat[1] = mx;
    }
    for (int p = tid; p < nb; p += nt) dvec[j0 + p] = dd[p];
    // the block's factor (unit diagonal implied; the diagonal keeps d)
    for (int idx = tid; idx < nb * nb; idx += nt) {
      const int r = idx % nb, c = idx / nb;
      if (r > c) S[(size_t)(j0 + c) * n + j0 + r] = A[r][c];
      else if (r == c) S[(size_t)(j0 + c) * n + j0 + r] = dd[c];
    }
    return;
  }
  // a panel row below the block: w = S[r, J] = x (L_JJ D)^T  =>  x_k d_k = w_k - sum_{q<k} (x_q d_q) L_JJ[k][q]
  // (x_q d_q kept in LDS, one column of 64 threads per q: rolled loops, no register pressure)
  __shared__ double xw[NB][64];
  const int r = j0 + nb + (blockIdx.x - 1) * 64 + tid;
  if (r >= n) return;  // (no barrier below)
  // the row's NB entries first (the stores below alias S for the compiler: fetched one per step of the
  // k loop they were a chain of 32 L2 round trips, 16 of the kernel's 29 us)
  double w[NB];
#pragma unroll
  for (int k = 0; k < NB; ++k) w[k] = (k < nb) ? S[(size_t)(j0 + k) * n + r] : 0.0;
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    if (k < nb) {
      double acc = w[k];
      for (int q = 0; q < k; ++q) acc = __builtin_fma(-xw[q][tid], A[k][q], acc);
      xw[k][tid] = acc;
      S[(size_t)(j0 + k) * n + r] = acc * di[k];  // L[r][j0 + k]
      W[(size_t)k * n + r] = acc;                 // W = L D
    }
  }
}

// grid: (tiles, tiles) over the trailing block rows/cols [j1, n); only tiles with tr >= tc work
__global__ void __launch_bounds__(256)
dense_ldl_update_kernel(double* __restrict__ S, int n, int j0, int j1, const double* __restrict__ W) {
  constexpr int NB = DENSE_NB, T = 32;
  const int tr = blockIdx.x, tc = blockIdx.y;
  if (tr < tc) return;
  __shared__ double Wt[NB][T + 1];  // W[r0 + i][k]
  __shared__ double Lt[NB][T + 1];  // L[c0 + j][k]
  const int r0 = j1 + tr * T, c0 = j1 + tc * T, tid = threadIdx.x;
  const int nb = j1 - j0;
  for (int idx = tid; idx < NB * T; idx += 256) {
    const int i = idx % T, k = idx / T;
    Wt[k][i] = (k < nb && r0 + i < n) ? W[(size_t)k * n + r0 + i] : 0.0;
    Lt[k][i] = (k < nb && c0 + i < n) ? S[(size_t)(j0 + k) * n + c0 + i] : 0.0;
  }
  __syncthreads();
  const int i = tid % T;
  for (int j = tid / T; j < T; j += 256 / T) {
    const int r = r0 + i, c = c0 + j;
    if (r < n && c < n && r >= c) {
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < NB; ++k) acc += Wt[k][i] * Lt[k][j];
      S[(size_t)c * n + r] -= acc;
    }
  }
}

// x = S^-1 b from the factors (L strictly lower in S, d in dvec); one workgroup, b and x in LDS
__global__ void __launch_bounds__(512)
dense_ldl_solve_kernel(const double* __restrict__ S, int n, const double* __restrict__ dvec,
                       const double* __restrict__ b, double b_sign, const double* __restrict__ b2, double* __restrict__ x) {
  extern __shared__ double y[];  // [n]
  constexpr int NB = DENSE_NB;
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int i = tid; i < n; i += nt) y[i] = (b2 ? b2[i] : 0.0) + b_sign * b[i];
  __syncthreads();
  double* red = y + n;        // [nt] reduction scratch
  double* Lb = red + nt;      // [NB][NB + 1] the panel's diagonal block (unit lower)
  auto stage_block = [&](int j0, int nb) {
    for (int idx = tid; idx < NB * NB; idx += nt) {
      const int r = idx % NB, c = idx / NB;
      Lb[r * (NB + 1) + c] = (r < nb && c < r) ? S[(size_t)(j0 + c) * n + j0 + r] : 0.0;
    }
  };
  // forward: L y = b, panel by panel
  for (int j0 = 0; j0 < n; j0 += NB) {
    const int nb = (n - j0 < NB) ? n - j0 : NB;
    stage_block(j0, nb);
    __syncthreads();
    if (tid < 64) {  // the NB x NB unit-lower triangle: one wavefront, lane r keeps y_r and row r of L
      const int r = tid & 31;
      double yr = (r < nb) ? y[j0 + r] : 0.0;
      double lr[NB];
#pragma unroll
      for (int p = 0; p < NB; ++p) lr[p] = Lb[r * (NB + 1) + p];
#pragma unroll
      for (int p = 0; p < NB; ++p) {
        const double yp = rdlane(yr, p);  // final: rows < p have been eliminated
        if (r > p) yr -= lr[p] * yp;
      }
      if (tid < nb) y[j0 + tid] = yr;
    }
    __syncthreads();
    for (int r = j0 + nb + tid; r < n; r += nt) {
      double lv[NB];   // (all loads in flight at once: a rolled loop was a chain of L2 round trips)
#pragma unroll
      for (int k = 0; k < NB; ++k) lv[k] = (k < nb) ? S[(size_t)(j0 + k) * n + r] : 0.0;
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < NB; ++k) acc = __builtin_fma(lv[k], y[j0 + (k < nb ? k : 0)], acc);
      y[r] -= acc;
    }
    __syncthreads();
  }
  for (int i = tid; i < n; i += nt) y[i] /= dvec[i];
  __syncthreads();
  // backward: L^T x = z, panels from the bottom
  const int last = ((n - 1) / NB) * NB;
  for (int j0 = last; j0 >= 0; j0 -= NB) {
    const int nb = (n - j0 < NB) ? n - j0 : NB;
    stage_block(j0, nb);
    // y_J -= L[below, J]^T x_below: a wavefront pair per column k sums over the rows (coalesced
    // along r), then the partial sums are reduced through LDS
    {
      const int lane = tid & 63, w = tid >> 6, nw = nt >> 6;
      for (int k = w; k < NB; k += nw) {
        double acc = 0.0;
        if (k < nb) {
          const int rb = j0 + nb + lane;
          double sv[8];   // (eight loads in flight; the rolled loop waited for each)
#pragma unroll
          for (int t = 0; t < 8; ++t) sv[t] = (rb + 64 * t < n) ? S[(size_t)(j0 + k) * n + rb + 64 * t] : 0.0;
#pragma unroll
          for (int t = 0; t < 8; ++t)
            if (rb + 64 * t < n) acc += sv[t] * y[rb + 64 * t];
          for (int r = rb + 512; r < n; r += 64) acc += S[(size_t)(j0 + k) * n + r] * y[r];
        }
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        if (lane == 0) red[k] = acc;
      }
      __syncthreads();
      if (tid < nb) y[j0 + tid] -= red[tid];
      __syncthreads();
    }
    if (tid < 64) {  // L_JJ^T x = z: lane c keeps x_c and column c of L (= row c of L^T)
      const int cc = tid & 31;
      double xc = (cc < nb) ? y[j0 + cc] : 0.0;
      double lc[NB];
#pragma unroll
      for (int p = 0; p < NB; ++p) lc[p] = Lb[p * (NB + 1) + cc];
#pragma unroll
      for (int p = NB - 1; p >= 0; --p) {
        const double xp = rdlane(xc, p);
        if (cc < p) xc -= lc[p] * xp;  // (rows >= nb of the staged block are zero)
      }
      if (tid < nb) y[j0 + tid] = xc;
    }
    __syncthreads();
  }
  for (int i = tid; i < n; i += nt) x[i] = y[i];
}

}  // namespace idto_dev
