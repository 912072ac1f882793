// penta_spd.h — fast block-Thomas factor+solve for the symmetric positive-definite
// block penta-diagonal Gauss-Newton Hessian (reference
// optimizer/penta_diagonal_solver.h:124-248, [Benkert & Fischer 2007]).
//
// Same recursion as the reference, specialised to the symmetric case
// (D_i = B_{i+1}^T, E_i = A_{i+2}^T  =>  K_{i+1} = H_i^T):
//   H_i  = B_{i+1}^T - H_{i-1}^T Z_{i-1}
//   S_i  = C_i - A_i Z_{i-2} - H_{i-1}^T Y_{i-1}                 (the reference's G_i)
//   [Y_i | Z_i | rt_i] = S_i^{-1} [H_i | A_{i+2}^T | r_i - A_i rt_{i-2} - H_{i-1}^T rt_{i-1}]
//   x_i  = rt_i - Y_i x_{i+1} - Z_i x_{i+2}
// What differs from the reference: S_i^{-1}[...] is formed by Gauss-Jordan
// elimination WITHOUT pivoting instead of Eigen::PartialPivLU — S_i is a Schur
// complement of an SPD matrix, hence SPD, and elimination without pivoting is
// backward stable for SPD matrices.  Results therefore agree with the pivoted LU
// to round-off (cond(H) * eps), not bit for bit; `penta_kernel` in kernels.h is
// the bit-exact restatement and stays available (IDTO_SOLVER_REFERENCE).
//
// Mapping to the hardware: the augmented matrix [S_i | H_i | E_i | r] lives in the
// REGISTERS of one wavefront per 64-k right-hand-side columns, one column per lane
// (k <= 32 rows = at most 64 VGPRs); a pivot step broadcasts the pivot column with
// v_readlane and is otherwise pure per-lane FMAs: no LDS traffic and no barrier
// inside the 19 dependent pivot steps.  The block products (H^T Z, H^T Y, A Z)
// between two eliminations are done by all waves from LDS.  One workgroup: the
// recursion over i is inherently sequential.
#pragma once

#include <hip/hip_runtime.h>

namespace idto_dev {

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// Gauss-Jordan on the columns held by this wavefront: lanes 0..k-1 hold the columns of
// the SPD block S, the other lanes right-hand-side columns; on exit those hold S^{-1} rhs.
template <int KMAX, bool EXACT>
__device__ __forceinline__ void gauss_jordan_wave(double (&xr)[KMAX], int k) {
#pragma unroll
  for (int j = 0; j < KMAX; ++j) {
    if (EXACT || j < k) {
      const double d = readlane_f64(xr[j], j);
      const double inv = 1.0 / d;
      const double t = xr[j] * inv;
      // the whole pivot column first (wave-uniform values, they live in SGPRs), then the
      // FMAs: one SGPR-hazard wait per pivot instead of one per row.  Row j+1 is updated
      // first so that the next pivot's reciprocal can overlap the remaining updates.
      double m[KMAX];
#pragma unroll
      for (int r = 0; r < KMAX; ++r) m[r] = (r != j && (EXACT || r < k)) ? readlane_f64(xr[r], j) : 0.0;
#pragma unroll
      for (int rr = 1; rr < KMAX; ++rr) {
        const int r = (j + rr) % KMAX;
        if (EXACT || r < k) xr[r] = __builtin_fma(-m[r], t, xr[r]);
      }
      xr[j] = t;
    }
  }
}

struct PentaSpdLds {
  // offsets (in doubles) into dynamic LDS; ks = odd column stride
  int W, Y0, Z0, H0, rt0, in0, bl, bl_size, end, kks, rts, kk;  // ring slot s of Y = Y0 + s * kks, ...
};
__host__ __device__ inline PentaSpdLds penta_spd_layout(int n, int k, int nrhs) {
  PentaSpdLds L;
  const int ks = k | 1, ncr = 2 * k + nrhs;
  int o = 0;
  L.W = o; o += (k + ncr) * ks;
  L.kks = k * ks; L.rts = nrhs * k; L.kk = k * k;
  L.Y0 = o; o += 2 * L.kks;
  L.Z0 = o; o += 3 * L.kks;
  L.H0 = o; o += 2 * L.kks;
  L.rt0 = o; o += 3 * L.rts;
  L.in0 = o; o += 4 * L.kk;
  L.bl = o;
  L.bl_size = (nrhs * n * k <= 4096) ? nrhs * n * k : 0;  // right-hand sides staged in LDS when small
  o += L.bl_size;
  L.end = o;
  return L;
}

// b: [nrhs][n*k] right-hand sides (rhs = rhs_sign * b), x: [nrhs][n*k] solutions.
// Yst, Zst: [n][k*k] (column-major blocks), kept for later solves / inspection.
template <int KMAX, int NT, bool EXACT>
__global__ void __launch_bounds__(NT)
penta_spd_kernel(int n, int k, const double* __restrict__ HA, const double* __restrict__ HB,
                 const double* __restrict__ HC, const double* __restrict__ b, double rhs_sign, int nrhs,
                 double* __restrict__ x, double* __restrict__ Yst, double* __restrict__ Zst) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6;
  const int kk = k * k, ks = k | 1, ncr = 2 * k + nrhs, per_wave = 64 - k;
  const int gj_waves = (ncr + per_wave - 1) / per_wave;
  const size_t nk = (size_t)n * k;
  const PentaSpdLds L = penta_spd_layout(n, k, nrhs);
  double* Wm = lds + L.W;

  for (int idx = tid; idx < L.bl; idx += nt) lds[idx] = 0.0;
  for (int idx = tid; idx < L.bl_size; idx += nt) lds[L.bl + idx] = rhs_sign * b[idx];

  // register prefetch of the next row's input blocks A_i, B_{i+1}, C_i, A_{i+2}
  constexpr int PMAX = 16;  // ceil(4 * 32*32 / 256)
  double pre[PMAX];
  auto fetch = [&](int i) {
#pragma unroll
    for (int s = 0; s < PMAX; ++s) {
      const int idx = tid + s * nt;
      double val = 0.0;
      if (idx < 4 * kk && i < n) {
        const int which = idx / kk, e = idx - which * kk;
        if (which == 0) val = HA[(size_t)i * kk + e];
        else if (which == 1) val = (i + 1 < n) ? HB[(size_t)(i + 1) * kk + e] : 0.0;
        else if (which == 2) val = HC[(size_t)i * kk + e];
        else val = (i + 2 < n) ? HA[(size_t)(i + 2) * kk + e] : 0.0;
      }
      pre[s] = val;
    }
  };
  fetch(0);
  __syncthreads();

  for (int i = 0; i < n; ++i) {
    double* Ai = lds + L.in0;
    double* Bn = Ai + kk;
    double* Ci = Bn + kk;
    double* An2 = Ci + kk;
    const double* Yp = lds + L.Y0 + ((i + 1) & 1) * L.kks;      // Y_{i-1}
    double* Yn = lds + L.Y0 + (i & 1) * L.kks;                  // Y_i
    const double* Zp = lds + L.Z0 + ((i + 2) % 3) * L.kks;      // Z_{i-1}
    const double* Zpp = lds + L.Z0 + ((i + 1) % 3) * L.kks;     // Z_{i-2}
    double* Zn = lds + L.Z0 + (i % 3) * L.kks;                  // Z_i
    const double* Hp = lds + L.H0 + ((i + 1) & 1) * L.kks;      // H_{i-1}
    double* Hn = lds + L.H0 + (i & 1) * L.kks;                  // H_i
    const double* rtp = lds + L.rt0 + ((i + 2) % 3) * L.rts;    // rt_{i-1}
    const double* rtpp = lds + L.rt0 + ((i + 1) % 3) * L.rts;   // rt_{i-2}
    double* rtn = lds + L.rt0 + (i % 3) * L.rts;                // rt_i

    // stage this row's inputs, start fetching the next row's
#pragma unroll
    for (int s = 0; s < PMAX; ++s) {
      const int idx = tid + s * nt;
      if (idx < 4 * kk) lds[L.in0 + idx] = pre[s];
    }
    fetch(i + 1);
    // write back the previous row's Y, Z (coalesced, fire and forget)
    if (i > 0) {
      for (int idx = tid; idx < kk; idx += nt) {
        const int c = idx / k, r = idx - c * k;
        Yst[(size_t)(i - 1) * kk + idx] = Yp[c * ks + r];
        Zst[(size_t)(i - 1) * kk + idx] = Zp[c * ks + r];
      }
    }
    __syncthreads();

    // ---- augmented matrix [S | H | E | r] (column-major, stride ks)
    if (EXACT) {
      // thread = (row r, column group): row r of A_i and of K_i = H_{i-1}^T stay in registers
      // and are reused for every column of the group; Y/Z/rt columns are LDS broadcasts.
      constexpr int K = KMAX;
      const int r = tid % K, cg = tid / K, ncg = nt / K;
      if (cg < ncg) {
        double arow[K], hrow[K];
#pragma unroll
        for (int m = 0; m < K; ++m) { arow[m] = Ai[m * K + r]; hrow[m] = Hp[r * ks + m]; }
        const int ncomp = 2 * K + nrhs;  // computed columns: S (K), H (K), rhs
        for (int cc = cg; cc < ncomp; cc += ncg) {
          double a0 = 0.0, a1 = 0.0;
          if (cc < K) {  // S_i
            const double* zc = Zpp + cc * ks;
            const double* yc = Yp + cc * ks;
#pragma unroll
            for (int m = 0; m < K; ++m) { a0 = __builtin_fma(arow[m], zc[m], a0); a1 = __builtin_fma(hrow[m], yc[m], a1); }
            Wm[cc * ks + r] = (Ci[cc * K + r] - a0) - a1;
          } else if (cc < 2 * K) {  // H_i
            const int c2 = cc - K;
            const double* zc = Zp + c2 * ks;
#pragma unroll
            for (int m = 0; m < K; ++m) a0 = __builtin_fma(hrow[m], zc[m], a0);
            const double val = Bn[r * K + c2] - a0;
            Wm[cc * ks + r] = val;
            Hn[c2 * ks + r] = val;
          } else {  // right-hand sides
            const int j = cc - 2 * K;
            const double* r2 = rtpp + j * K;
            const double* r1 = rtp + j * K;
#pragma unroll
            for (int m = 0; m < K; ++m) { a0 = __builtin_fma(arow[m], r2[m], a0); a1 = __builtin_fma(hrow[m], r1[m], a1); }
            const double bval = L.bl_size ? lds[L.bl + j * (int)nk + i * K + r]
                                          : rhs_sign * b[(size_t)j * nk + (size_t)i * K + r];
            Wm[(3 * K + j) * ks + r] = (bval - a0) - a1;
          }
        }
        // E_i = A_{i+2}^T
        for (int cc = cg; cc < K; cc += ncg) Wm[(2 * K + cc) * ks + r] = An2[r * K + cc];
      }
    } else {
    for (int idx = tid; idx < (k + ncr) * k; idx += nt) {
      const int c = idx / k, r = idx - c * k;
      double val;
      if (c < k) {  // S_i
        double acc = Ci[c * k + r];
#pragma unroll 4
        for (int m = 0; m < k; ++m) acc = __builtin_fma(-Ai[m * k + r], Zpp[c * ks + m], acc);
#pragma unroll 4
        for (int m = 0; m < k; ++m) acc = __builtin_fma(-Hp[r * ks + m], Yp[c * ks + m], acc);
        val = acc;
      } else if (c < 2 * k) {  // H_i
        const int cc = c - k;
        double acc = Bn[r * k + cc];  // B_{i+1}^T
#pragma unroll 4
        for (int m = 0; m < k; ++m) acc = __builtin_fma(-Hp[r * ks + m], Zp[cc * ks + m], acc);
        val = acc;
        Hn[cc * ks + r] = acc;
      } else if (c < 3 * k) {  // E_i = A_{i+2}^T
        val = An2[r * k + (c - 2 * k)];
      } else {  // right-hand sides
        const int j = c - 3 * k;
        double acc = L.bl_size ? lds[L.bl + j * (int)nk + i * k + r]
                               : rhs_sign * b[(size_t)j * nk + (size_t)i * k + r];
#pragma unroll 4
        for (int m = 0; m < k; ++m) acc = __builtin_fma(-Ai[m * k + r], rtpp[j * k + m], acc);
#pragma unroll 4
        for (int m = 0; m < k; ++m) acc = __builtin_fma(-Hp[r * ks + m], rtp[j * k + m], acc);
        val = acc;
      }
      Wm[c * ks + r] = val;
    }
    }
    __syncthreads();

    // ---- Gauss-Jordan in registers, one wavefront per (64 - k) right-hand-side columns
    if (wave < gj_waves) {
      const int rc = wave * per_wave + (lane - k);  // rhs column of this lane (lanes >= k)
      const bool is_rhs = lane >= k && rc < ncr;
      const int col = (lane < k) ? lane : (is_rhs ? k + rc : 0);
      double xr[KMAX];
#pragma unroll
      for (int r = 0; r < KMAX; ++r) xr[r] = (r < k) ? Wm[col * ks + r] : 0.0;
      gauss_jordan_wave<KMAX, EXACT>(xr, k);
      if (is_rhs) {
        if (rc < k) {
#pragma unroll
          for (int r = 0; r < KMAX; ++r) if (r < k) Yn[rc * ks + r] = xr[r];
        } else if (rc < 2 * k) {
#pragma unroll
          for (int r = 0; r < KMAX; ++r) if (r < k) Zn[(rc - k) * ks + r] = xr[r];
        } else {
          const int j = rc - 2 * k;
#pragma unroll
          for (int r = 0; r < KMAX; ++r)
            if (r < k) {
              rtn[j * k + r] = xr[r];
              x[(size_t)j * nk + (size_t)i * k + r] = xr[r];  // rt_i parked in x until the backward pass
            }
        }
      }
    }
    __syncthreads();
  }
  {  // last row's Y, Z
    const double* Yl = lds + L.Y0 + ((n - 1) & 1) * L.kks;
    const double* Zl = lds + L.Z0 + ((n - 1) % 3) * L.kks;
    for (int idx = tid; idx < kk; idx += nt) {
      const int c = idx / k, r = idx - c * k;
      Yst[(size_t)(n - 1) * kk + idx] = Yl[c * ks + r];
      Zst[(size_t)(n - 1) * kk + idx] = Zl[c * ks + r];
    }
  }

  // ---- backward substitution: x_i = rt_i - Y_i x_{i+1} - Z_i x_{i+2}
  // Y_i, Z_i are re-read from LDS-friendly HBM blocks with a one-row register prefetch;
  // x_{i+1}, x_{i+2} live in the rt ring (x_{n-1} = rt_{n-1} is already in slot (n-1)%3).
  double* Yb = lds + L.in0;  // staging for Y_i (kk)
  double* Zb = Yb + kk;
  constexpr int QMAX = 4;  // ceil(32*32 / 256)
  double py[QMAX], pz[QMAX];
  auto fetch_yz = [&](int i) {
#pragma unroll
    for (int s = 0; s < QMAX; ++s) {
      const int idx = tid + s * nt;
      const bool ok = idx < kk && i >= 0;
      py[s] = ok ? Yst[(size_t)i * kk + idx] : 0.0;
      pz[s] = ok ? Zst[(size_t)i * kk + idx] : 0.0;
    }
  };
  __syncthreads();  // Yst/Zst of all rows written by this block: make them visible to its own loads
  __threadfence_block();
  fetch_yz(n - 2);
  for (int i = n - 2; i >= 0; --i) {
#pragma unroll
    for (int s = 0; s < QMAX; ++s) {
      const int idx = tid + s * nt;
      if (idx < kk) { Yb[idx] = py[s]; Zb[idx] = pz[s]; }
    }
    fetch_yz(i - 1);
    __syncthreads();
    const double* x1 = lds + L.rt0 + ((i + 1) % 3) * L.rts;
    const double* x2 = lds + L.rt0 + ((i + 2) % 3) * L.rts;
    double* xi = lds + L.rt0 + (i % 3) * L.rts;
    // wave w handles right-hand sides j = w, w + nwaves, ...; lane = row + 32 * half
    const int nwaves = nt >> 6;
    for (int j = wave; j < nrhs; j += nwaves) {
      const int r = lane & 31, half = lane >> 5;
      double acc = 0.0;
      if (r < k) {
        const double* Mb = half ? Zb : Yb;
        const double* xv = (half ? x2 : x1) + j * k;
        const bool use = half ? (i + 2 < n) : true;
        if (use)
          for (int m = 0; m < k; ++m) acc = __builtin_fma(Mb[m * k + r], xv[m], acc);
      }
      acc += __shfl_xor(acc, 32);
      if (half == 0 && r < k) {
        const double val = x[(size_t)j * nk + (size_t)i * k + r] - acc;
        xi[j * k + r] = val;
        x[(size_t)j * nk + (size_t)i * k + r] = val;
      }
    }
    __syncthreads();
  }
}

}  // namespace idto_dev
