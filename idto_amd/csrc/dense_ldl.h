// dense_ldl.h — S lambda = r for the dense symmetric positive definite Schur complement
// S = J H^-1 J^T of the Lagrange multipliers (reference optimizer/trajectory_optimizer.cc:1395:
// Eigen ldlt() of an n_eq x n_eq matrix, n_eq = 40 .. 360 for the example models), on the device
// so that S (up to 1 MB) never crosses PCIe and the 8 MFLOP factorisation does not sit on one
// host core.
//
// Right-looking blocked LDL^T without pivoting, panels of NB = 32 columns, S column-major with
// only the lower triangle referenced:
//   dense_ldl_step_kernel    (one launch per panel, a workgroup per 32 x 32 tile of the trailing lower triangle)
//                            factorises the NB x NB diagonal block (every workgroup for itself), solves the panel
//                            rows of its row and column block, subtracts W_R L_C^T from its tile on the matrix
//                            cores; L goes to a matrix of its own, the pivots and their extremes to dvec / stat;
//   dense_ldl_solve_kernel   (1 workgroup)  L y = r, D z = y, L^T x = z, blocked the same way.
// No pivoting: S is positive definite whenever the constraint Jacobian has full row rank.  The
// pivot extremes are returned; the caller falls back to the host's pivoted LDL^T (which, like
// Eigen's, tolerates semi-definite S) when min pivot <= 1e-13 * max pivot or a pivot is not finite.
#pragma once
#include <hip/hip_runtime.h>

#include "penta_ldl.h"  // rdlane

namespace idto_dev {

constexpr int DENSE_NB = 32;

// ---- one panel step in ONE launch.  stat[0] = min pivot so far, stat[1] = max pivot so far (initialised by the
// caller: +inf, 0).  (Rounds 1-2 ran a panel kernel - diagonal block + row solves, 21 us - and an update kernel -
// 6 us - per panel: two launches, W = L D through HBM.)
// grid (tiles, tiles) over the 32 x 32 tiles of the trailing block [j1, n) x [j1, n), tiles with tr >= tc work;
// (1, 1) for the last panel.  Every workgroup factorises the NB x NB diagonal block for itself (wavefront 0,
// registers + v_readlane), solves the panel rows of ITS row block and ITS column
// block against it (wavefronts 0 and 1, one thread per row: W = L D for the rows, L for the columns) and subtracts
// W_R L_C^T from its tile with v_mfma_f64_16x16x4 (four wavefronts, one 16 x 16 quadrant each, 8 k-steps).  The
// panel of S is only READ here - L goes to a matrix of its own (`Lm`, same shape; the tiles of the first tile column
// write their rows, tile (0, 0) the diagonal block, d and the pivot statistics) - so no workgroup can see a panel
// another one has already overwritten.
__global__ void __launch_bounds__(256)
dense_ldl_step_kernel(double* __restrict__ S, double* __restrict__ Lm, int n, int j0, double* __restrict__ dvec,
                      double* __restrict__ stat, const double* __restrict__ b, double b_sign, const double* __restrict__ b2,
                      double* __restrict__ rv, double* __restrict__ yv) {
  constexpr int NB = DENSE_NB, T = 32;
  const int tr = blockIdx.x, tc = blockIdx.y;
  if (tr < tc) return;
  __shared__ double A[NB][NB + 1];     // diagonal block, then its unit-lower factor
  __shared__ double dd[NB], di[NB];
  __shared__ double xw[2][NB][T + 1];  // [0]: W[r0 + i][k] = x_k d_k of the row block, [1]: of the column block
  __shared__ double Lt[NB][T + 1];     // L[c0 + j][k]
  __shared__ double ys[NB];            // forward substitution of the right-hand side: y of this panel
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nb = (n - j0 < NB) ? n - j0 : NB, j1 = j0 + nb;
  const int r0 = j1 + tr * T, c0 = j1 + tc * T;
  for (int idx = tid; idx < NB * NB; idx += 256) {
    const int r = idx % NB, c = idx / NB;
    A[r][c] = (r < nb && c < nb && r >= c) ? S[(size_t)(j0 + c) * n + j0 + r] : 0.0;
  }
  // the panel rows of both blocks, fetched while the diagonal block is factorised (wavefronts 1, 2)
  double w[NB];
  const int which = wave - 1;                                   // 0: row block, 1: column block
  const int prow = (which == 0 ? r0 : c0) + (lane & 31);
  const bool solver = (wave == 1 || (wave == 2 && tr != tc)) && lane < 32 && r0 < n;
  if (solver) {
#pragma unroll
    for (int k = 0; k < NB; ++k) w[k] = (k < nb && prow < n) ? S[(size_t)(j0 + k) * n + prow] : 0.0;
  }
  // the right-hand side rides along (L y = r panel by panel: the solve kernel then only substitutes backwards):
  // rv holds r with the panels before this one eliminated (rows >= j0; a panel's own rows are never written again), yv
  // the finished y; the first step forms r = b2 + b_sign b itself
  const bool rhs = rv != nullptr && tc == 0 && wave == 3 && lane < 32;
  double rj = 0.0, rr_blk = 0.0;
  if (rhs) {
    const int gj = j0 + lane, gr = r0 + lane;
    if (lane < nb) rj = (j0 == 0) ? (b2 ? b2[gj] : 0.0) + b_sign * b[gj] : rv[gj];
    if (gr < n) rr_blk = (j0 == 0) ? (b2 ? b2[gr] : 0.0) + b_sign * b[gr] : rv[gr];
  }
  __syncthreads();
  if (wave == 0) {
    const int rr = lane & 31;
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
    if (lane < NB) {
#pragma unroll
      for (int c = 0; c < NB; ++c) A[rr][c] = a[c];
    }
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    if (lane < NB) {
      const double d = A[lane][lane];
      dd[lane] = d;
      di[lane] = fast_rcp(d);
    }
  }
  __syncthreads();
  if (tr == 0 && tc == 0) {   // the diagonal block's factor, its pivots, the statistics
    if (tid == 0) {
      double mn = stat[0], mx = stat[1];
      for (int p = 0; p < nb; ++p) {
        const double d = dd[p];
        if (!(d > mn)) mn = d;  // also catches NaN
        if (d > mx) mx = d;
      }
      stat[0] = mn; stat[1] = mx;
    }
    for (int p = tid; p < nb; p += 256) dvec[j0 + p] = dd[p];
    for (int idx = tid; idx < nb * nb; idx += 256) {
      const int r = idx % nb, c = idx / nb;
      if (r > c) Lm[(size_t)(j0 + c) * n + j0 + r] = A[r][c];
      else if (r == c) Lm[(size_t)(j0 + c) * n + j0 + r] = dd[c];
    }
  }
  if (rv != nullptr && tc == 0 && wave == 3) {   // y_J = L_JJ^-1 r_J: lane r keeps r_r and row r of the unit-lower factor
    const int r = lane & 31;
    double lr[NB];
#pragma unroll
    for (int p = 0; p < NB; ++p) lr[p] = A[r][p];
#pragma unroll
    for (int p = 0; p < NB; ++p) {
      const double yp = rdlane(rj, p);  // final: rows < p have been eliminated
      if (r > p) rj -= lr[p] * yp;
    }
    if (lane < NB) ys[lane] = (lane < nb) ? rj : 0.0;
    // (y goes to a vector of its own: every first-column workgroup of this launch reads r_J = rv[j0 ..] when it starts,
    // and nothing orders those reads against this store - written in place, a workgroup that started late substituted
    // an already substituted vector)
    if (tr == 0 && lane < nb) yv[j0 + lane] = rj;
  }
  if (r0 >= n) return;   // (the last panel: nothing below it)
  // a panel row below the block: w = S[r, J] = x (L_JJ D)^T  =>  x_k d_k = w_k - sum_{q<k} (x_q d_q) L_JJ[k][q]
  if (solver) {
    const int i = lane & 31;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      double acc = 0.0;
      if (k < nb) {
        acc = w[k];
        for (int q = 0; q < k; ++q) acc = __builtin_fma(-xw[which][q][i], A[k][q], acc);
      }
      xw[which][k][i] = acc;
      const double l = acc * di[k < nb ? k : 0];
      if (which == 1 || tr == tc) Lt[k][i] = (k < nb) ? l : 0.0;
      if (which == 0 && tc == 0 && k < nb && prow < n) Lm[(size_t)(j0 + k) * n + prow] = l;   // L[r][j0 + k]
    }
  }
  __syncthreads();
  if (rhs && r0 + lane < n) {   // r_R -= L_R,J y_J  (L = W / d)
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < NB; ++k) acc = __builtin_fma(xw[0][k][lane] * di[k], ys[k], acc);
    rv[r0 + lane] = rr_blk - acc;
  }
  // tile -= W_R L_C^T on the matrix cores: wavefront q takes the 16 x 16 quadrant (q >> 1, q & 1)
  {
    using d4 = __attribute__((ext_vector_type(4))) double;
    const int qr = wave >> 1, qc = wave & 1, fl = lane & 15, fk = lane >> 4;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int sq = 0; sq < NB / 4; ++sq) {
      const int kr = 4 * sq + fk;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xw[0][kr][16 * qr + fl], Lt[kr][16 * qc + fl], acc, 0, 0, 0);
    }
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int r = r0 + 16 * qr + fk + 4 * rg, c = c0 + 16 * qc + fl;
      if (r < n && c < n && r >= c) S[(size_t)c * n + r] -= acc[rg];
    }
  }
}

// H as a dense matrix (reference PentaDiagonalMatrix::MakeDense, optimizer/penta_diagonal_matrix.cc:107-140, which
// SolveLinearSystemInPlace's kDenseLdlt branch factorises: optimizer/trajectory_optimizer.cc:2088-2093): the lower
// triangle of the n x n column-major matrix from the lower bands A (two block rows below the diagonal), B (one), C
// (diagonal; its lower triangle is read), blocks column-major bs x bs, block row t at t * bs * bs.  One workgroup per
// block column; entries outside the band are zeroed (the factorisation fills the whole trailing triangle).
__global__ void __launch_bounds__(256)
dense_from_bands_kernel(const double* __restrict__ A, const double* __restrict__ B, const double* __restrict__ C, int nblk,
                        int bs, double* __restrict__ S) {
  const int s = blockIdx.x, n = nblk * bs, rows = n - s * bs;
  for (int idx = threadIdx.x; idx < rows * bs; idx += blockDim.x) {
    const int c = idx / rows, rr = idx % rows;          // column c of block column s, row s * bs + rr
    const int dt = rr / bs, r = rr % bs, t = s + dt;
    double v = 0.0;
    if (dt == 0) v = (r >= c) ? C[(size_t)t * bs * bs + (size_t)c * bs + r] : 0.0;
    else if (dt == 1) v = B[(size_t)t * bs * bs + (size_t)c * bs + r];
    else if (dt == 2) v = A[(size_t)t * bs * bs + (size_t)c * bs + r];
    S[(size_t)(s * bs + c) * n + s * bs + rr] = v;
  }
}

// x = S^-1 b from the factors (L strictly lower in S, d in dvec); one workgroup, b and x in LDS
__global__ void __launch_bounds__(512)
dense_ldl_solve_kernel(const double* __restrict__ S, int n, const double* __restrict__ dvec,
                       const double* __restrict__ b, double b_sign, const double* __restrict__ b2, double* __restrict__ x,
                       int forward_done) {   // forward_done: b already is L^-1 r (dense_ldl_step_kernel carried it along)
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
  for (int j0 = 0; j0 < n && !forward_done; j0 += NB) {
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
