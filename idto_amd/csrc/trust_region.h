// trust_region.h — the O(num_vars) bookkeeping of one trust-region iteration on the resident
// arrays, so that the host reads back a handful of scalars per iteration instead
// of g and the three Hessian bands:
//   tr_prepare_kernel   CalcScaleFactors (optimizer/trajectory_optimizer.cc:1225-1255), the scaled
//                       merit gradient g~ = D (g + J^T lambda) (:1204-1223, :1435-1456), the products
//                       H~ g~ and H~ w with H~ = D H D (:1181-1202, penta_diagonal_matrix.cc:181-207)
//                       where w = D^-1 H^-1 (g + J^T lambda), and every inner product CalcDoglegPoint
//                       (:2108-2202) and CalcTrustRatio (:1979-2035) need:
//                       out = [g~.g~, g~.H~g~, w.w, g~.w, g~.H~w, w.H~w, q.q, h.h, h.lambda]
//   tr_trial_kernel     dq = D (a g~ + b w) - every branch of the dogleg is such a combination -,
//                       q_trial = q + dq (optionally with normalised quaternions, :2691-2707),
//                       out = [dq.dq, g~.D^-1 dq ... ] (see the kernel)
// Single workgroup each: num_vars is a few hundred to ~1.4k.
#pragma once

#include <hip/hip_runtime.h>

namespace idto_dev {

// block-wide sums of NS values per thread; result valid in thread 0
template <int NS>
__device__ __forceinline__ void block_sums(double (&val)[NS], double* scratch /* [NS * 16] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    double x = val[s];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_down(x, off);
    if (lane == 0) scratch[s * 16 + wave] = x;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      double x = 0.0;
      for (int w = 0; w < nw; ++w) x += scratch[s * 16 + w];
      val[s] = x;
    }
  }
  __syncthreads();
}

// (H x)[i*K + r] for the symmetric block penta-diagonal H given by its lower bands (blocks column-major)
__device__ __forceinline__ double penta_row(const double* __restrict__ HA, const double* __restrict__ HB,
                                            const double* __restrict__ HC, const double* __restrict__ x, int nblk, int K,
                                            int i, int r) {
  const int kk = K * K;
  double acc = 0.0;
  if (i >= 2) { const double* M = HA + (size_t)i * kk; const double* xs = x + (i - 2) * K; for (int c = 0; c < K; ++c) acc += M[c * K + r] * xs[c]; }
  if (i >= 1) { const double* M = HB + (size_t)i * kk; const double* xs = x + (i - 1) * K; for (int c = 0; c < K; ++c) acc += M[c * K + r] * xs[c]; }
  { const double* M = HC + (size_t)i * kk; const double* xs = x + i * K; for (int c = 0; c < K; ++c) acc += M[c * K + r] * xs[c]; }
  if (i + 1 < nblk) { const double* M = HB + (size_t)(i + 1) * kk; const double* xs = x + (i + 1) * K; for (int c = 0; c < K; ++c) acc += M[r * K + c] * xs[c]; }
  if (i + 2 < nblk) { const double* M = HA + (size_t)(i + 2) * kk; const double* xs = x + (i + 2) * K; for (int c = 0; c < K; ++c) acc += M[r * K + c] * xs[c]; }
  return acc;
}

// scale factor of variable (block i, row r) from the Hessian's diagonal (TO.cc:1239-1252)
__device__ __forceinline__ double scale_factor(int scaling_method, double hii, double prev) {
  switch (scaling_method) {
    case 0: return __builtin_fmin(1.0, 1.0 / __builtin_sqrt(hii));
    case 1: return __builtin_fmin(prev, 1.0 / __builtin_sqrt(hii));
    case 2: return __builtin_fmin(1.0, 1.0 / __builtin_sqrt(__builtin_sqrt(hii)));
    case 3: return __builtin_fmin(prev, 1.0 / __builtin_sqrt(__builtin_sqrt(hii)));
    default: return 1.0;
  }
}

// scaling_method: -1 none, else ScalingMethod (solver_parameters.h:52-62): 0 kSqrt, 1 kAdaptiveSqrt,
// 2 kDoubleSqrt, 3 kAdaptiveDoubleSqrt.  y = ysign * yin is H^-1 (g + J^T lambda) (unscaled H).
// Grid: one workgroup per block row i (a single workgroup walking all rows took 28 us: the two
// band products are a few hundred dependent global loads per thread).  The workgroup forms D, g~,
// w of its own rows and of the four neighbouring block rows it multiplies with (D from the
// diagonal of H: nothing is exchanged between workgroups; for the adaptive methods the previous D
// of the neighbours is read before anybody overwrites it - `Dprev` is a separate array that
// tr_prepare_sum_kernel refreshes afterwards), then the rows of H~ g~ and H~ w with one thread per
// (row, band block) and a sum over the five band blocks in LDS, and writes its nine partial sums;
// tr_prepare_sum_kernel adds the partial sums in block order (deterministic).
__global__ void __launch_bounds__(256)
tr_prepare_rows_kernel(int nblk, int K, const double* __restrict__ HA, const double* __restrict__ HB,
                       const double* __restrict__ HC, const double* __restrict__ g, const double* __restrict__ jtl,
                       const double* __restrict__ yin, double ysign, const double* __restrict__ q, int scaling_method,
                       const double* __restrict__ Dprev, double* __restrict__ D, double* __restrict__ gt,
                       double* __restrict__ w, const double* __restrict__ slab, int slab_stride, int tau_off,
                       const int* __restrict__ dofs, int nu, int N, const double* __restrict__ lambda,
                       double* __restrict__ partial) {
  extern __shared__ double lds[];
  const int i = blockIdx.x, tid = threadIdx.x, nt = blockDim.x, kk = K * K;
  double* xt = lds;            // [5 K]  D g~ of block rows i-2 .. i+2
  double* xy = xt + 5 * K;     // [5 K]  y
  double* dl = xy + 5 * K;     // [K]    D of this block row
  double* pt = dl + K;         // [5 K]  partial products of H (D g~), per band block
  double* py = pt + 5 * K;     // [5 K]  ... of H y
  double* gl = py + 5 * K;     // [K]    g~ of this block row
  double* wl = gl + K;         // [K]    w
  double* scratch = wl + K;
  for (int idx = tid; idx < 5 * K; idx += nt) {
    const int j = idx / K, r = idx - j * K, bi = i - 2 + j;
    double vt = 0.0, vy = 0.0;
    if (bi >= 0 && bi < nblk) {
      const int v = bi * K + r;
      const double d = (scaling_method >= 0) ? scale_factor(scaling_method, HC[(size_t)bi * kk + r * K + r], Dprev[v]) : 1.0;
      const double gm = jtl ? g[v] + jtl[v] : g[v];
      const double gti = d * gm, yi = ysign * yin[v];
      vt = d * gti; vy = yi;
      if (j == 2) { const double wi = yi / d; dl[r] = d; gl[r] = gti; wl[r] = wi; D[v] = d; gt[v] = gti; w[v] = wi; }
    }
    xt[idx] = vt; xy[idx] = vy;
  }
  __syncthreads();
  // band block j multiplies x_{i-2+j}: A_i, B_i, C_i, B_{i+1}^T, A_{i+2}^T (blocks column-major)
  for (int idx = tid; idx < 5 * K; idx += nt) {
    const int j = idx / K, r = idx - j * K, bi = i - 2 + j;
    double at = 0.0, ay = 0.0;
    if (bi >= 0 && bi < nblk) {
      const double* M = (j == 0) ? HA + (size_t)i * kk : (j == 1) ? HB + (size_t)i * kk : (j == 2) ? HC + (size_t)i * kk
                      : (j == 3) ? HB + (size_t)(i + 1) * kk : HA + (size_t)(i + 2) * kk;
      const int sr = (j <= 2) ? 1 : K, sc = (j <= 2) ? K : 1;   // M(r, c) or M(c, r)
      const double* a = xt + j * K;
      const double* y = xy + j * K;
      for (int c = 0; c < K; ++c) { const double m = M[r * sr + c * sc]; at += m * a[c]; ay += m * y[c]; }
    }
    pt[idx] = at; py[idx] = ay;
  }
  __syncthreads();
  double s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (tid < K) {
    const int r = tid, v = i * K + r;
    const double d = dl[r];
    const double Hg = d * ((((pt[r] + pt[K + r]) + pt[2 * K + r]) + pt[3 * K + r]) + pt[4 * K + r]);
    const double Hw = d * ((((py[r] + py[K + r]) + py[2 * K + r]) + py[3 * K + r]) + py[4 * K + r]);
    const double gti = gl[r], wi = wl[r], qi = q[v];
    s[0] = gti * gti; s[1] = gti * Hg; s[2] = wi * wi; s[3] = gti * wi; s[4] = gti * Hw; s[5] = wi * Hw; s[6] = qi * qi;
  }
  if (nu > 0 && i < N) {   // h = tau_i[unactuated] (TO.cc:1274-1278); lambda only when the constraints are enforced
    for (int j = tid; j < nu; j += nt) {
      const double h = slab[(size_t)i * slab_stride + tau_off + dofs[j]];
      s[7] += h * h;
      if (lambda) s[8] += h * lambda[i * nu + j];
    }
  }
  block_sums<9>(s, scratch);
  if (tid == 0)
    for (int k = 0; k < 9; ++k) partial[i * 9 + k] = s[k];
}

__global__ void tr_prepare_sum_kernel(int nblk, const double* __restrict__ partial, double* __restrict__ out,
                                      const double* __restrict__ D, double* __restrict__ Dprev, int n) {
  const int tid = threadIdx.x;
  if (tid < 9) {
    double acc = 0.0;
    for (int i = 0; i < nblk; ++i) acc += partial[i * 9 + tid];
    out[tid] = acc;
  }
  for (int idx = tid; idx < n; idx += blockDim.x) Dprev[idx] = D[idx];   // the adaptive methods' memory
}

// dq = D (a g~ + b w), q_trial = q + dq; out = [dq.dq, g~.(a g~ + b w)] ; quaternions of q_trial
// normalised on request (quat[] = their start indices within one time step)
__global__ void __launch_bounds__(1024)
tr_trial_kernel(int n, int nq, const double* __restrict__ D, const double* __restrict__ gt, const double* __restrict__ w,
                double a, double b, int scaling, const double* __restrict__ q, double* __restrict__ q_trial,
                double* __restrict__ dq_out, const int* __restrict__ quat, int nquat, double* __restrict__ out) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x, nt = blockDim.x;
  double s[2] = {0, 0};
  for (int idx = tid; idx < n; idx += nt) {
    const double dqs = a * gt[idx] + b * w[idx];
    const double dq = scaling ? D[idx] * dqs : dqs;
    dq_out[idx] = dq;
    q_trial[idx] = q[idx] + dq;
    s[0] += dq * dq;
    s[1] += gt[idx] * dqs;
  }
  if (nquat > 0) {
    __syncthreads();
    const int nsteps = n / nq;
    for (int idx = tid; idx < nsteps * nquat; idx += nt) {
      const int t = idx / nquat, qs = quat[idx - t * nquat];
      double* qq = q_trial + (size_t)t * nq + qs;
      const double nrm = __builtin_sqrt(qq[0] * qq[0] + qq[1] * qq[1] + qq[2] * qq[2] + qq[3] * qq[3]);
      for (int k = 0; k < 4; ++k) qq[k] /= nrm;
    }
  }
  block_sums<2>(s, lds);
  if (tid == 0) { out[0] = s[0]; out[1] = s[1]; }
}

// h(q_trial) . lambda  (the merit function at the trial point uses the multipliers of the current
// iterate, TO.cc:2010-2016); one workgroup
__global__ void tr_hlambda_kernel(const double* __restrict__ slab, int slab_stride, int tau_off,
                                  const int* __restrict__ dofs, int nu, int N, const double* __restrict__ lambda,
                                  double* __restrict__ out) {
  extern __shared__ double lds[];
  double s[1] = {0};
  for (int idx = threadIdx.x; idx < nu * N; idx += blockDim.x) {
    const int t = idx / nu, j = idx - t * nu;
    s[0] += slab[(size_t)t * slab_stride + tau_off + dofs[j]] * lambda[idx];
  }
  block_sums<1>(s, lds);
  if (threadIdx.x == 0) out[0] = s[0];
}

}  // namespace idto_dev
