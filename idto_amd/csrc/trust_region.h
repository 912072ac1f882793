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

#include "batch.h"

namespace idto_dev {

// Measurement build only (-DIDTO_TR_STAMPS, tools/tr_stamps.py): wall-clock stamps of the trust-region kernels' phases
#ifdef IDTO_TR_STAMPS
__device__ unsigned long long g_tr_stamps[64];
#define TR_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.y == 0) g_tr_stamps[i] = wall_clock64(); } while (0)
#define TR_STAMP_B0(i) do { if (blockIdx.x == 0) TR_STAMP(i); } while (0)
#else
#define TR_STAMP(i) do { } while (0)
#define TR_STAMP_B0(i) do { } while (0)
#endif

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
// slab record k: [dtau_k/dq_{k-1} | dtau_k/dq_k | dtau_k/dq_{k+1} | tau_k], blocks nv x nq stored column by column:
// J[(t, dof), col] for a global column index col in [0, (N+1) nq)  (constraints.h jac_entry - that header comes later)
__device__ __forceinline__ double tr_jac_entry(const double* __restrict__ slab, int slab_stride, int nq, int nv, int t,
                                               int dof, int N, int col) {
  const int tc = col / nq, i = col - tc * nq;
  const int which = tc - t + 1;  // 0: q_{t-1}, 1: q_t, 2: q_{t+1}
  if (which < 0 || which > 2) return 0.0;
  if ((which == 0 && t < 2) || (which == 1 && t < 1)) return 0.0;  // q_0 is not a variable of tau_t's rows (TO.cc:1316-1322)
  return slab[(size_t)t * slab_stride + (size_t)which * nv * nq + i * nv + dof];
}

// The banded KKT step (kkt.h) taken apart where it is used: z_t = [-w_t ; lambda_{t-1}] is the solver's x.  With .z set,
// tr_prepare_rows_body forms H^-1 (g + J^T lambda) = -z and J^T lambda of the five block rows it stages itself -
// kkt_extract_kernel's expressions, same bits - and writes its own block row's to w_out / jtl_out / lambda_out for
// the kernels that follow (a launch of its own for that was 4.6 us of every constrained iteration).
struct TrKkt {
  const double* z;     // null: off (yin, jtl, lambda are read)
  int KK, nv;          // nq + nu; the model's nv
  double *w_out, *jtl_out, *lambda_out;
};

struct TrRowsArgs {
  int nblk, K;
  const double *HA, *HB, *HC, *g, *jtl, *yin;
  double ysign;
  const double* q;
  int scaling_method;
  const double* Dprev;
  double *D, *gt, *w;
  const double* slab;
  int slab_stride, tau_off;
  const int* dofs;
  int nu, N;
  const double* lambda;
  double* partial;        // [nblk][9] (stepwise calls: tr_prepare_sum_kernel adds them)
  const double* freeze;   // (resident loop) the sticky flags word: once it is non-zero D, g~, w keep the values of the last decided iteration
  // resident loop (tr_iter_kernel): ten sums per block row - the tenth is (g + J^T lambda) . dq_old over the row, for the
  // convergence criteria - handed to every workgroup of the launch with the launch's epoch in every word (ll_store)
  double* part_ll;        // [nblk][TR_NSUM][2], or null: `partial` is written
  unsigned epoch;
  const double* dq_old;   // the step that led to this iterate, or null
  TrKkt kx;
};
constexpr int TR_NSUM = 10;
// LDS of tr_prepare_rows_body, in doubles (nt = 256): its arrays, then the scratch of the sums
__host__ __device__ constexpr int tr_rows_lds(int K) { return 24 * K + TR_NSUM * 32; }

// (penta_nd.h has the same pair; this header comes first)
__device__ __forceinline__ void tr_ll_store(double* slot2, double v, unsigned epoch) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v), e = (unsigned long long)epoch << 32;
  unsigned long long* q = reinterpret_cast<unsigned long long*>(slot2);
  __hip_atomic_store(q, (b & 0xffffffffull) | e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(q + 1, (b >> 32) | e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool tr_ll_try(const double* slot2, unsigned epoch, double& v) {
  const unsigned long long* q = reinterpret_cast<const unsigned long long*>(slot2);
  const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v = __longlong_as_double((long long)((a & 0xffffffffull) | (b << 32)));
  return (unsigned)(a >> 32) == epoch && (unsigned)(b >> 32) == epoch;
}

// U: how many entries of a band block's row a thread keeps in registers (K <= U and 5 K <= blockDim.x: every
// configuration of the five examples and their KKT systems); wider blocks take the loop that loads as it multiplies.
template <int U>
__device__ __forceinline__ void tr_prepare_rows_body(const TrRowsArgs& A, double* lds) {
  const int nblk = A.nblk, K = A.K, scaling_method = A.scaling_method, slab_stride = A.slab_stride, tau_off = A.tau_off;
  const int nu = A.nu, N = A.N;
  const double* __restrict__ HA = A.HA; const double* __restrict__ HB = A.HB; const double* __restrict__ HC = A.HC;
  const double* __restrict__ g = A.g; const double* __restrict__ jtl = A.jtl; const double* __restrict__ yin = A.yin;
  const double ysign = A.ysign;
  const double* __restrict__ q = A.q; const double* __restrict__ Dprev = A.Dprev;
  double* __restrict__ D = A.D; double* __restrict__ gt = A.gt; double* __restrict__ w = A.w;
  const double* __restrict__ slab = A.slab; const int* __restrict__ dofs = A.dofs;
  const double* __restrict__ lambda = A.lambda; double* __restrict__ partial = A.partial;
  const int i = blockIdx.x, tid = threadIdx.x, nt = blockDim.x, kk = K * K;
  const bool frozen = A.freeze && *A.freeze != 0.0;
  double* xt = lds;            // [5 K]  D g~ of block rows i-2 .. i+2
  double* xy = xt + 5 * K;     // [5 K]  y
  double* dl = xy + 5 * K;     // [K]    D of this block row
  double* pt = dl + K;         // [5 K]  partial products of H (D g~), per band block
  double* py = pt + 5 * K;     // [5 K]  ... of H y
  double* gl = py + 5 * K;     // [K]    g~ of this block row
  double* wl = gl + K;         // [K]    w
  double* gml = wl + K;        // [K]    g + J^T lambda
  double* scratch = gml + K;   // (tr_rows_lds)
  // The band blocks' rows and q of this block row have addresses that depend on nothing computed here, and everything
  // this launch reads was written by the launch before it - cold in this XCD's L2.  Requested first, they share the
  // round trip of D, g~, w below (as the products' own loads behind the barrier they were a second cold round trip
  // of 2 us, and q a third).
  const bool pre = K <= U && 5 * K <= nt;
  double m[U];
#pragma unroll
  for (int u = 0; u < U; ++u) m[u] = 0.0;
  if (pre && tid < 5 * K) {
    const int j = tid / K, r = tid - j * K, bi = i - 2 + j;
    if (bi >= 0 && bi < nblk) {
      const double* M = (j == 0) ? HA + (size_t)i * kk : (j == 1) ? HB + (size_t)i * kk : (j == 2) ? HC + (size_t)i * kk
                      : (j == 3) ? HB + (size_t)(i + 1) * kk : HA + (size_t)(i + 2) * kk;
      const int sr = (j <= 2) ? 1 : K, sc = (j <= 2) ? K : 1;   // M(r, c) or M(c, r)
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (u < K) m[u] = M[r * sr + u * sc];
    }
  }
  const double qi_pre = (tid < K) ? q[i * K + tid] : 0.0;
  const double dqo_pre = (tid < K && A.dq_old) ? A.dq_old[i * K + tid] : 0.0;
  for (int idx = tid; idx < 5 * K; idx += nt) {
    const int j = idx / K, r = idx - j * K, bi = i - 2 + j;
    double vt = 0.0, vy = 0.0;
    if (bi >= 0 && bi < nblk) {
      const int v = bi * K + r;
      const double d = (scaling_method >= 0) ? scale_factor(scaling_method, HC[(size_t)bi * kk + r * K + r], Dprev[v]) : 1.0;
      double jt = 0.0, yv;
      if (A.kx.z) {   // (kkt_extract_kernel's sums: ascending time step, then dof)
        for (int s = (bi >= 1 ? bi - 1 : 0); s <= bi + 1 && s < N; ++s)
          for (int jj = 0; jj < nu; ++jj)
            jt += tr_jac_entry(slab, slab_stride, K, A.kx.nv, s, dofs[jj], N, v) * A.kx.z[(size_t)(s + 1) * A.kx.KK + K + jj];
        yv = -A.kx.z[(size_t)bi * A.kx.KK + r];
        if (j == 2) { A.kx.w_out[v] = yv; A.kx.jtl_out[v] = jt; }
      } else {
        if (jtl) jt = jtl[v];
        yv = yin[v];
      }
      const double gm = (jtl || A.kx.z) ? g[v] + jt : g[v];
      const double gti = d * gm, yi = ysign * yv;
      vt = d * gti; vy = yi;
      if (j == 2) {
        const double wi = yi / d;
        dl[r] = d; gl[r] = gti; wl[r] = wi; gml[r] = gm;
        // (the resident loop's workgroups write these once they know that the iteration counts: tr_iter_kernel)
        if (!frozen && !A.part_ll) { D[v] = d; gt[v] = gti; w[v] = wi; }
      }
    }
    xt[idx] = vt; xy[idx] = vy;
  }
  __syncthreads();
  TR_STAMP_B0(1);
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
      if (pre) {   // (idx == tid: the row this thread fetched above; the sums keep their order)
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (u < K) { at += m[u] * a[u]; ay += m[u] * y[u]; }
      } else
      // (eight loads in flight at a time: the rolled loop with one load per step was a chain of L2 round trips)
      for (int c0 = 0; c0 < K; c0 += 8) {
        double m8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) m8[u] = (c0 + u < K) ? M[r * sr + (c0 + u) * sc] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (c0 + u < K) { at += m8[u] * a[c0 + u]; ay += m8[u] * y[c0 + u]; }
      }
    }
    pt[idx] = at; py[idx] = ay;
  }
  __syncthreads();
  TR_STAMP_B0(2);
  double s[TR_NSUM] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (tid < K) {
    const int r = tid, v = i * K + r;
    const double d = dl[r];
    const double Hg = d * ((((pt[r] + pt[K + r]) + pt[2 * K + r]) + pt[3 * K + r]) + pt[4 * K + r]);
    const double Hw = d * ((((py[r] + py[K + r]) + py[2 * K + r]) + py[3 * K + r]) + py[4 * K + r]);
    const double gti = gl[r], wi = wl[r], qi = qi_pre;
    s[0] = gti * gti; s[1] = gti * Hg; s[2] = wi * wi; s[3] = gti * wi; s[4] = gti * Hw; s[5] = wi * Hw; s[6] = qi * qi;
    s[9] = gml[r] * dqo_pre;
  }
  if (nu > 0 && i < N) {   // h = tau_i[unactuated] (TO.cc:1274-1278); lambda only when the constraints are enforced
    for (int j = tid; j < nu; j += nt) {
      const double h = slab[(size_t)i * slab_stride + tau_off + dofs[j]];
      s[7] += h * h;
      if (A.kx.z) s[8] += h * A.kx.z[(size_t)(i + 1) * A.kx.KK + K + j];   // lambda_i = the multiplier rows of z_{i+1}
      else if (lambda) s[8] += h * lambda[i * nu + j];
    }
  }
  if (K <= 32 && nu <= 32) {
    // Only lanes < 32 of the first wavefront hold anything but 0.0.  block_sums' result for them - the shuffle tree
    // x_l += x_{l+32}, += x_{l+16}, ... += x_{l+1}, then 0.0 + the wavefronts' results - by one thread per sum from LDS,
    // same association, same bits (adding the other lanes' and wavefronts' +0.0 changes nothing: a sum that starts
    // with `+ 0.0` is never -0.0).  The nine trees of 54 dependent ds_bpermute round trips were 4 us of this kernel.
    double* V = scratch;   // [TR_NSUM][32]
    if (tid < 32) {
#pragma unroll
      for (int k = 0; k < TR_NSUM; ++k) V[k * 32 + tid] = s[k];
    }
    __syncthreads();
    if (tid < TR_NSUM) {
      double x[32];
#pragma unroll
      for (int l = 0; l < 32; ++l) x[l] = V[tid * 32 + l] + 0.0;
#pragma unroll
      for (int off = 16; off >= 1; off >>= 1)
#pragma unroll
        for (int l = 0; l < off; ++l) x[l] += x[l + off];
      const double tot = 0.0 + x[0];
      if (A.part_ll) tr_ll_store(A.part_ll + 2 * (i * TR_NSUM + tid), tot, A.epoch);
      else if (tid < 9) partial[i * 9 + tid] = tot;
    }
  } else {
    block_sums<TR_NSUM>(s, scratch);
    if (tid == 0)
      for (int k = 0; k < TR_NSUM; ++k) {
        if (A.part_ll) tr_ll_store(A.part_ll + 2 * (i * TR_NSUM + k), s[k], A.epoch);
        else if (k < 9) partial[i * 9 + k] = s[k];
      }
  }
  TR_STAMP_B0(3);
}

__device__ __forceinline__ void tr_prepare_rows(const TrRowsArgs& A, double* lds) {
  if (A.K <= 4) tr_prepare_rows_body<4>(A, lds);
  else if (A.K <= 8) tr_prepare_rows_body<8>(A, lds);
  else if (A.K <= 20) tr_prepare_rows_body<20>(A, lds);
  else tr_prepare_rows_body<32>(A, lds);
}

__global__ void __launch_bounds__(256) tr_prepare_rows_kernel(TrRowsArgs A) {
  extern __shared__ double lds[];
  tr_prepare_rows(A, lds);
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
  extern __shared__ double lds[];   // [2 n] the terms, [2 nsteps] their sums per time step
  const int tid = threadIdx.x, nt = blockDim.x, nsteps = n / nq;
  double* x0 = lds;
  double* x1 = lds + n;
  double* r0 = lds + 2 * n;
  double* r1 = r0 + nsteps;
  for (int idx = tid; idx < n; idx += nt) {
    const double dqs = a * gt[idx] + b * w[idx];
    const double dq = scaling ? D[idx] * dqs : dqs;
    dq_out[idx] = dq;
    q_trial[idx] = q[idx] + dq;
    x0[idx] = dq * dq;
    x1[idx] = gt[idx] * dqs;
  }
  __syncthreads();
  if (nquat > 0) {
    for (int idx = tid; idx < nsteps * nquat; idx += nt) {
      const int t = idx / nquat, qs = quat[idx - t * nquat];
      double* qq = q_trial + (size_t)t * nq + qs;
      const double nrm = __builtin_sqrt(qq[0] * qq[0] + qq[1] * qq[1] + qq[2] * qq[2] + qq[3] * qq[3]);
      for (int k = 0; k < 4; ++k) qq[k] /= nrm;
    }
  }
  // the two sums in the order of the resident loop (tr_iter_kernel: a time step's terms in order by its workgroup,
  // cost_kernel: the time steps in order)
  for (int t = tid; t < nsteps; t += nt) {
    double s0 = 0.0, s1 = 0.0;
    for (int r = 0; r < nq; ++r) { s0 += x0[t * nq + r]; s1 += x1[t * nq + r]; }
    r0[t] = s0; r1[t] = s1;
  }
  __syncthreads();
  if (tid == 0) {
    double s0 = 0.0, s1 = 0.0;
    for (int t = 0; t < nsteps; ++t) { s0 += r0[t]; s1 += r1[t]; }
    out[0] = s0; out[1] = s1;
  }
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


// ---------------------------------------------------------------------------
// The trust-region iteration without the host (idto_hip_tr_solve): tr_iter_kernel = tr_prepare_rows +
// (last workgroup to finish) tr_prepare_sum + CalcDoglegPoint + tr_trial; the bookkeeping that
// follows the cost of the trial point - CalcTrustRatio, accept / reject, the radius update
// (TO.cc:2004-2034, :2550-2553, :2614-2622) - is tr_decide(), called by the one workgroup of
// cost_kernel.  The inverse dynamics at the trial point are evaluated together with their partials
// (fd_kernel mode >= 1): an accepted trial point - the common case - is the next iterate and its
// assembly follows at once; after a rejected one the gated assembly leaves g and H alone.
// The state that travels from launch to launch lives in device memory:
enum {
  TRS_DELTA = 0,   // trust-region radius
  TRS_COST,        // L(q)
  TRS_A, TRS_B,    // the step of this iteration: dq = D (a g~ + b w)
  TRS_ACTIVE,      // the trust-region constraint is active (:2160-2199)
  TRS_FLAGS,       // (as an integer value) TRF_*: sticky, the remaining iterations are idle
  TRS_ITER,        // iterations decided so far
  TRS_ACCEPTED,    // the last decision: gates the assembly of the next iteration (a rejected step keeps g, H)
  TRS_PREVCOST,    // convergence check: L of the iterate the last accepted step started from (VerifyConvergenceCriteria's previous_cost)
  TRS_CUR,         // which of the two sets of fd_kernel outputs holds the iterate's (batch.h AltSel)
  TRS_CHECK,       // an accepted step's convergence criteria are to be evaluated by the next tr_iter_kernel (it needs g there)
  TRS_DQN,         // |dq| of that step
  TRS_COUNT = 12
};
static_assert(TRS_CUR == IDTO_TRS_CUR, "batch.h and trust_region.h disagree");
enum { TRF_DOGLEG = 1, TRF_NONFINITE = 2, TRF_NOT_DESCENT = 4, TRF_SINGULAR_S = 8 /* constraints.h constraint_lambda_kernel */,
       TRF_CONVERGED = 16 /* a convergence criterion held after an accepted step (TO.cc:2654-2689): not an error */,
       TRF_FACTORIZATION = 32 /* the factorisation behind this iteration's step met a bad pivot (penta_diagonal_solver.h:181-185) */,
       TRF_SOLVER_TIMEOUT = 64 /* a wait between the solver's workgroups ran out in the launch behind this iteration's step: the step is
                                  garbage, nothing is built on it (penta_ldl.h spin_wait; the host repeats the solve) */ };
// one row of per-iteration statistics (TrajectoryOptimizerStats::push_data, TO.cc:2586-2598)
enum { TRR_COST = 0, TRR_DELTA, TRR_RHO, TRR_QNORM, TRR_DQNORM, TRR_DQHNORM, TRR_GNORM, TRR_DLDQ, TRR_HNORM,
       TRR_ACCEPTED, TRR_CLOCK /* wall_clock64 ticks (100 MHz) */, TRR_A, TRR_B, TRR_COST_TRIAL, TRR_FLAGS,
       TRR_MERIT /* L(q) + h(q).lambda (TO.cc:2583; = the cost without enforced constraints) */,
       TRR_REASON /* ConvergenceReason bitmask of the accepted step (trajectory_optimizer_solution.h:17-23), 0: none / not checked */,
       TRR_COUNT = 17 };

// VerifyConvergenceCriteria (TO.cc:2654-2689) inside the resident loop.  The criteria of the step accepted in
// iteration k need the gradient AT the new iterate: they are evaluated by the tr_iter_kernel of iteration k + 1
// (or by one more, check-only, launch after the last iteration) from
//   |L_prev - L| < abs_cost + rel_cost L,   |g.dq| < abs_grad + rel_grad L,   |dq| < abs_state + rel_state |q|
// and written into row k; a satisfied criterion sets the sticky TRF_CONVERGED: the remaining iterations idle.
struct TrConvergence {
  int on, check_only;
  double rel_cost, abs_cost, rel_grad, abs_grad, rel_state, abs_state;
  double* rows;   // [iterations][TRR_COUNT]
};

struct TrIterArgs {
  TrRowsArgs rows;               // (.part_ll, .epoch, .dq_old: the hand-over of the ten sums per block row)
  AltSel alt;                    // rows.slab is set A's: h = tau[unactuated] is read from the iterate's set
  double* part2;                 // [nblk][2] dq.dq and g~.(a g~ + b w) of each block row: cost_kernel adds them in block order
  double* out;                   // [11] the nine inner products (then, by cost_kernel, dq.dq and g~.D^-1 dq)
  double* state;                 // [TRS_COUNT]
  int n, nq, scaling, nquat;
  const int* quat;
  double* q_trial;
  double* dq;
  TrConvergence conv;
  const unsigned* fact_status;   // the solver's status word (host-mapped) and the id of the factorisation this iteration's step
  unsigned fact_id;              // came from: a failure in the MIDDLE of the resident loop is flagged in its own iteration
  unsigned* timeout_status;      // ... and the word a launch whose waits between workgroups ran out writes its id to
  size_t pstride, rows_stride;   // batch contexts: grid.y = problem (arena stride in bytes; doubles between the problems' rows)
  // (rows.kx.z set) the KKT factorisation's 1 / d - solver row i (= block row i + first_row), lane r at Dinv[i dstride + r] -
  // for the multiplier pivots' range (TRF_SINGULAR_S), and the KKT context's arena stride
  const double* kdinv; int kdstride, kfirst_row;
  size_t kstride;
  double* curpre;       // (or null) a copy of the current-set word as THIS iteration finds it: the deciding solver launch's
                        // assembly locates the trial point's records by it while the word itself may already have been flipped
  int debug_skip_row;   // test aid (option "debug_skip_role" 100 + i): block row i's workgroup returns at once - a partner that is not resident
};

// (TO.cc:2204-2242 SolveDoglegQuadratic; *ok = false where the reference throws)
__device__ inline double tr_dogleg_quadratic(double a, double b, double c, bool* ok) {
  if (!(a > 0)) { *ok = false; return 0.0; }
  double s;
  if (a < 2.220446049250313e-16) {
    s = -c / b;
  } else {
    const double bt = b / a, ct = c / a;
    const double det = bt * bt - 4 * ct;
    if (!(det > 0)) { *ok = false; return 0.0; }
    s = (-bt + __builtin_sqrt(det)) / 2;
  }
  if (!(0 < s && s < 1)) *ok = false;
  return s;
}

// The launch's extra workgroup (tr_iter_kernel: block nblk; gn_small.h's folded iteration: block 1).  Its three words go
// behind the block rows' sums: slots TR_NSUM nblk, + 1, + 2 of part_ll.
__device__ __forceinline__ void tr_status_reader(const TrIterArgs& T, int nblk, int K, int tid) {
  // The launch's extra workgroup.  The solver's status words live in host-mapped memory: a read is a round trip over
  // PCIe, 3.7 us - and every barrier of a workgroup stands behind the loads its wavefronts have in flight (the old last
  // workgroup spent its "sums" phase there).  One thread reads them for the problem, at the launch's start, and hands
  // them to the block rows' workgroups with their sums; it is back before the rows are.
  if (tid == 0) {
    unsigned fact_word = 0u, timeout_word = 0u;
    if (T.fact_status && T.timeout_status == T.fact_status + 2) {   // (a single problem: the four words are adjacent, one read)
      typedef unsigned u4 __attribute__((ext_vector_type(4)));
      const u4 wds = *reinterpret_cast<const volatile u4*>(T.fact_status);
      fact_word = wds[0]; timeout_word = wds[2];
    } else {
      if (T.fact_status) fact_word = __hip_atomic_load(T.fact_status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (T.timeout_status) timeout_word = __hip_atomic_load(T.timeout_status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    tr_ll_store(T.rows.part_ll + 2 * (TR_NSUM * nblk), (double)fact_word, T.rows.epoch);
    tr_ll_store(T.rows.part_ll + 2 * (TR_NSUM * nblk + 1), (double)timeout_word, T.rows.epoch);
  }
  if (tid < 64) {
    // (KKT step) the multiplier pivots - 1 / d is what the factorisation keeps -: negative, finite, and min |d| / max |d| >
    // 1e-13; anything else = redundant constraints, the host's pivoted factorisation takes over (kkt_extract_kernel's
    // criterion, constraint_lambda_kernel's on the pivots of S)
    bool bad = false;
    if (T.rows.kx.z) {
      const int nu = T.rows.nu, N = T.rows.N;
      double imn = __builtin_inf(), imx = 0.0;
      bool finite = true;
      for (int idx = tid; idx < N * nu; idx += 64) {
        const int bt = 1 + idx / nu, j = idx - (bt - 1) * nu;
        const double iv = -T.kdinv[(size_t)(bt - T.kfirst_row) * T.kdstride + K + j];
        finite = finite && __builtin_isfinite(iv) && iv > 0.0;
        imn = __builtin_fmin(imn, iv); imx = __builtin_fmax(imx, iv);
      }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        imn = __builtin_fmin(imn, __shfl_xor(imn, off)); imx = __builtin_fmax(imx, __shfl_xor(imx, off));
      }
      bad = __builtin_amdgcn_ballot_w64(!finite) != 0ull || !(imn > 1e-13 * imx);   // (|d|: max = 1 / imn, min = 1 / imx)
    }
    if (tid == 0) tr_ll_store(T.rows.part_ll + 2 * (TR_NSUM * nblk + 2), bad ? 1.0 : 0.0, T.rows.epoch);
  }
}

// Thread 0 of a workgroup that holds the sums of all block rows S[TR_NSUM] and the loop's state words st[] as the launch
// found them: the convergence criteria of the step the previous iteration accepted (TrConvergence: g.dq with the merit
// function's gradient at THIS iterate and the dq that led here, S[9]), then - unless the pass only checks -
// CalcDoglegPoint normalised by Delta: pU = cU g~ (TO.cc:2157), pH = -w / Delta (:2139-2149).  `write`: this workgroup
// writes the state and the statistics; extra: TRF_* bits the caller found (factorisation, timeout, singular system).
// ab = [a, b, flags]; st[] is brought up to date.
__device__ inline void tr_conv_dogleg(const TrIterArgs& T, const double* S, double (&st)[TRS_COUNT], bool write, int extra, double* ab) {
  if (T.conv.on) {
    int flags = (int)st[TRS_FLAGS];
    if (st[TRS_CHECK] != 0.0 && (flags & ~TRF_CONVERGED) == 0) {
      const double gdq = S[9];
      const double cost = st[TRS_COST], prev = st[TRS_PREVCOST];
      int reason = 0;
      if (__builtin_fabs(prev - cost) < T.conv.abs_cost + T.conv.rel_cost * cost) reason |= 1;
      if (__builtin_fabs(gdq) < T.conv.abs_grad + T.conv.rel_grad * cost) reason |= 2;
      if (st[TRS_DQN] < T.conv.abs_state + T.conv.rel_state * __builtin_sqrt(S[6])) reason |= 4;
      const int k_prev = (int)st[TRS_ITER] - 1;
      if (write && k_prev >= 0) T.conv.rows[(size_t)k_prev * TRR_COUNT + TRR_REASON] = (double)reason;
      if (reason) st[TRS_FLAGS] = (double)(flags | TRF_CONVERGED);
    }
    st[TRS_CHECK] = 0.0;
    if (write) { T.state[TRS_FLAGS] = st[TRS_FLAGS]; T.state[TRS_CHECK] = 0.0; }
    if (T.conv.check_only) return;
  }
  const double gg = S[0], gHg = S[1], ww = S[2], gw = S[3];
  const double Delta = st[TRS_DELTA];
  int flags = (int)st[TRS_FLAGS] | extra;
  const double cU = -(gg / gHg) / Delta;
  const double pUn = __builtin_fabs(cU) * __builtin_sqrt(gg), pHn = __builtin_sqrt(ww) / Delta;
  double a, b, active;
  if (1.0 <= pUn) {          // :2160-2168
    a = (Delta / pUn) * cU; b = 0.0; active = 1.0;
  } else if (1.0 >= pHn) {   // :2171-2178
    a = 0.0; b = -1.0; active = 0.0;
  } else {                   // :2180-2199
    const double pUpU = cU * cU * gg, pHpH = ww / (Delta * Delta), pUpH = -cU * gw / Delta;
    bool ok = true;
    const double sq = tr_dogleg_quadratic(pHpH - 2 * pUpH + pUpU, 2 * (pUpH - pUpU), pUpU - 1.0, &ok);
    if (!ok) flags |= TRF_DOGLEG;
    a = Delta * (1.0 - sq) * cU; b = -sq; active = 1.0;
  }
  if (!(__builtin_isfinite(a) && __builtin_isfinite(b))) flags |= TRF_NONFINITE;
  st[TRS_A] = a; st[TRS_B] = b; st[TRS_ACTIVE] = active; st[TRS_FLAGS] = (double)flags;
  if (write) { T.state[TRS_A] = a; T.state[TRS_B] = b; T.state[TRS_ACTIVE] = active; T.state[TRS_FLAGS] = (double)flags; }
  ab[0] = a; ab[1] = b; ab[2] = (double)flags;
}

// One workgroup per block row (= time step), and every workgroup goes all the way:
//   1. its rows of D, g~, w, H~ g~, H~ w and their ten sums (tr_prepare_rows), published with the launch's epoch in every
//      word;
//   2. it polls the sums of ALL block rows (the data itself: one memory round trip behind the last writer, no counter,
//      no fence, no cache invalidation), adds them in block order and evaluates the convergence criteria and the dogleg -
//      every workgroup the same scalars from the same bits; workgroup 0 alone writes them to the state and the statistics;
//   3. dq and the trial point of ITS rows from the D, g~, w, q it still holds, and the row's dq.dq, g~.dqs.
// (Until round 6 the last workgroup to arrive did 2 and 3 for everybody: behind its acquire it fetched g~, w, D, q of
// the whole trajectory cold - 200 cache lines through one CU's miss queue, 5 of this kernel's 13.5 us at cheetah's size,
// 8 of 18.5 at allegro's.)  The workgroups of a problem wait for each other, so they must be resident together: N + 1 <=
// a few hundred workgroups of 256 threads, dispatched in order - in a batch the problems behind the dispatch front are
// complete and finish, so the front moves.  The wait is bounded (spin_wait's clock): on expiry the loop idles with
// TRF_SOLVER_TIMEOUT.
__global__ void __launch_bounds__(256) tr_iter_kernel(TrIterArgs T) {
  extern __shared__ double lds[];
  __shared__ double ab[3];
  __shared__ int timed_out;
  {   // problem of the batch: every array of the loop lives in the problem's arena
    const size_t o = (size_t)blockIdx.y * T.pstride;
    TrRowsArgs& A = T.rows;
    A.HA = at_problem(A.HA, o); A.HB = at_problem(A.HB, o); A.HC = at_problem(A.HC, o); A.g = at_problem(A.g, o);
    if (A.jtl) A.jtl = at_problem(A.jtl, o);
    A.yin = at_problem(A.yin, o); A.q = at_problem(A.q, o); A.Dprev = at_problem(A.Dprev, o); A.D = at_problem(A.D, o);
    A.gt = at_problem(A.gt, o); A.w = at_problem(A.w, o); A.part_ll = at_problem(A.part_ll, o);
    if (A.lambda) A.lambda = at_problem(A.lambda, o);
    if (A.freeze) A.freeze = at_problem(A.freeze, o);
    A.slab = at_problem(A.slab, o + (size_t)alt_offset(T.alt, o));   // (h = tau[unactuated] is read from the iterate's set)
    T.part2 = at_problem(T.part2, o); T.out = at_problem(T.out, o); T.state = at_problem(T.state, o);
    if (T.curpre) T.curpre = at_problem(T.curpre, o);
    T.q_trial = at_problem(T.q_trial, o); T.dq = at_problem(T.dq, o);
    A.dq_old = T.conv.on ? T.dq : nullptr;
    if (A.kx.z) {
      const size_t ok = (size_t)blockIdx.y * T.kstride;
      A.kx.z = at_problem(A.kx.z, ok); T.kdinv = at_problem(T.kdinv, ok);
      A.kx.w_out = at_problem(A.kx.w_out, o); A.kx.jtl_out = at_problem(A.kx.jtl_out, o); A.kx.lambda_out = at_problem(A.kx.lambda_out, o);
    }
    T.conv.rows += (size_t)blockIdx.y * T.rows_stride;
    if (T.fact_status) T.fact_status += 2 * blockIdx.y;
  }
#ifdef IDTO_TR_STAMPS
  if (T.conv.check_only) return;   // (measurement build: the stamps of the last DECIDING iteration stay)
#endif
  TR_STAMP_B0(0);
  const int tid = threadIdx.x, nt = blockDim.x, nblk = T.rows.nblk, K = T.rows.K, i = blockIdx.x;
  const bool first = blockIdx.x == 0;
  if (i == nblk) { tr_status_reader(T, nblk, K, tid); return; }
  if (i == T.debug_skip_row) return;
  // The loop's state words: nobody writes them before every workgroup has published its sums (workgroup 0, after its
  // poll below), and a wavefront's loads return in order - these are back before its first store of step 1.
  double st[TRS_COUNT];
#pragma unroll
  for (int k = 0; k < TRS_COUNT; ++k) st[k] = (tid == 0) ? T.state[k] : 0.0;
  const bool frozen = T.rows.freeze && *T.rows.freeze != 0.0;
  const double qi = (tid < K) ? T.rows.q[i * K + tid] : 0.0;
  // ---- 1. this block row
  tr_prepare_rows(T.rows, lds);
  TR_STAMP_B0(3);
  const double* dl = lds + 10 * K;   // (tr_prepare_rows_body's arrays: D, g~, w of this block row)
  const double* gl = lds + 21 * K;
  const double* wl = lds + 22 * K;
  double* part = lds + tr_rows_lds(K);       // [TR_NSUM nblk + 3]: the block rows' sums, the solver's two status words, "singular"
  double* S = part + TR_NSUM * nblk + 3;     // [TR_NSUM]
  double* rowx = S + TR_NSUM;                // [3 K] dq.dq, g~.dqs terms of the row; its trial point
  // ---- 2. the sums of every block row: polled where they are written, four per thread in flight
  {
    const int cnt = TR_NSUM * nblk + 3;
    bool fine = true;
    for (int base = 0; base < cnt; base += 4 * nt) {
      double v[4] = {0.0, 0.0, 0.0, 0.0};
      unsigned pending = 0u;
#pragma unroll
      for (int u = 0; u < 4; ++u) pending |= (base + tid + u * nt < cnt) ? (1u << u) : 0u;
      unsigned polls = 0;
      long long t0 = 0;
      while (pending) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (pending & (1u << u)) {
            double x;
            if (tr_ll_try(T.rows.part_ll + 2 * (base + tid + u * nt), T.rows.epoch, x)) { v[u] = x; pending &= ~(1u << u); }
          }
        if (!pending) break;
        __builtin_amdgcn_s_sleep(1);
        if (((++polls) & 1023u) == 0) {   // (bounded like the solvers' waits: penta_ldl.h spin_wait, 50 ms)
          const long long now = (long long)wall_clock64();
          if (t0 == 0) t0 = now;
          else if (now - t0 > 5000000) {
            if (T.timeout_status) {
              // (the count's upper half: THIS kernel's wait, not a solver's - the host then leaves the solvers as they are and
              // takes the loop that returns to it twice an iteration, which has no wait between workgroups: idto_hip.hip FactorStatus)
              __hip_atomic_store(T.timeout_status, T.fact_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
              __hip_atomic_fetch_add(T.timeout_status + 1, 0x10000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            fine = false;
            break;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int idx = base + tid + u * nt; if (idx < cnt) part[idx] = v[u]; }
    }
    if (tid == 0) timed_out = 0;
    __syncthreads();
    if (!fine) timed_out = 1;
  }
  __syncthreads();
  TR_STAMP_B0(4);
  if (tid < TR_NSUM) {   // in block order (the LDS reads eight at a time ahead of the chain of adds)
    double acc = 0.0;
    int b = 0;
    for (; b + 8 <= nblk; b += 8) {
      double t8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t8[u] = part[(b + u) * TR_NSUM + tid];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += t8[u];
    }
    for (; b < nblk; ++b) acc += part[b * TR_NSUM + tid];
    S[tid] = acc;
    if (first && tid < 9) T.out[tid] = acc;
  }
  // D, g~, w of this block row and the adaptive methods' memory (every workgroup has read its neighbours' by now) - unless
  // the loop idles already, or this iteration's multipliers came from a singular system
  const bool singular = part[TR_NSUM * nblk + 2] != 0.0;
  if (tid < K && !frozen && !singular) {
    const int v = i * K + tid;
    T.rows.D[v] = dl[tid]; T.rows.gt[v] = gl[tid]; T.rows.w[v] = wl[tid];
    const_cast<double*>(T.rows.Dprev)[v] = dl[tid];
  }
  if (T.rows.kx.z && i >= 1 && tid < T.rows.nu)   // lambda_{i-1}: the multiplier rows of z_i
    T.rows.kx.lambda_out[(size_t)(i - 1) * T.rows.nu + tid] = T.rows.kx.z[(size_t)i * T.rows.kx.KK + K + tid];
  __syncthreads();
  TR_STAMP_B0(5);
  // ---- the convergence criteria of the step the previous iteration accepted, then the dogleg (tr_conv_dogleg)
  if (tid == 0) {
    int extra = 0;
    if (T.fact_status && (unsigned)part[TR_NSUM * nblk] == T.fact_id) extra |= TRF_FACTORIZATION;
    if ((T.timeout_status && (unsigned)part[TR_NSUM * nblk + 1] == T.fact_id) || timed_out) extra |= TRF_SOLVER_TIMEOUT;
    if (singular) extra |= TRF_SINGULAR_S;
    tr_conv_dogleg(T, S, st, first, extra, ab);
    if (first && T.curpre) *T.curpre = st[TRS_CUR];
  }
  if (T.conv.on && T.conv.check_only) return;
  TR_STAMP_B0(6);
  __syncthreads();
  TR_STAMP_B0(7);
  // ---- 3. dq = D (a g~ + b w), q_trial = q + dq of this block row; the row's dq.dq and g~.(a g~ + b w), added in row
  // order (tr_trial_kernel: the same order)
  // (once a sticky flag is set - converged, or an error - the loop idles: dq and the trial point keep the values of the
  // last decided iteration, which is what the warm start hands on: TO.cc:2361-2385)
  const double a = ab[0], b = ab[1];
  const bool idle = ab[2] != 0.0;
  if (tid < K) {
    const double dqs = a * gl[tid] + b * wl[tid];
    const double dq = T.scaling ? dl[tid] * dqs : dqs;
    if (!idle) T.dq[i * K + tid] = dq;
    rowx[tid] = dq * dq; rowx[K + tid] = gl[tid] * dqs; rowx[2 * K + tid] = qi + dq;
  }
  __syncthreads();
  if (T.nquat > 0 && tid < T.nquat && i < T.n / T.nq) {   // (quaternions of this time step, normalised where they stand)
    double* qq = rowx + 2 * K + T.quat[tid];
    const double nrm = __builtin_sqrt(qq[0] * qq[0] + qq[1] * qq[1] + qq[2] * qq[2] + qq[3] * qq[3]);
    for (int k = 0; k < 4; ++k) qq[k] /= nrm;
  }
  if (tid == 64) {
    double x0 = 0.0, x1 = 0.0;
    for (int r = 0; r < K; ++r) { x0 += rowx[r]; x1 += rowx[K + r]; }
    T.part2[2 * i] = x0; T.part2[2 * i + 1] = x1;
  }
  if (T.nquat > 0) __syncthreads();
  if (tid < K && !idle) T.q_trial[i * K + tid] = rowx[2 * K + tid];
  TR_STAMP_B0(9);
}

struct TrDecideArgs {
  double* state;        // [TRS_COUNT]
  double* out;          // [11]: [0..8] from tr_iter_kernel; [9], [10] = dq.dq, g~.D^-1 dq are added up here from
  const double* part2;  // ... tr_iter_kernel's [nblk][2] per block row, in block order
  int nblk;
  double* rows;         // [iterations][TRR_COUNT]
  size_t rows_stride;   // doubles between the rows of consecutive problems of a batch
  double* q;            // the iterate, overwritten by q_trial when the step is accepted
  const double* q_trial;
  int n;
  double eta, Delta_max, eps;
  // enforced equality constraints (nu > 0): the merit function L + h.lambda at both points (TO.cc:1979-2003)
  const double* lambda;   // [N nu] multipliers of the iterate
  const int* dofs;        // [nu] unactuated degrees of freedom
  int nu, N, slab_stride, tau_off;
};

// thread 0 of cost_kernel's workgroup, with the cost of the trial point (and h(q + dq).lambda when
// constraints are enforced); returns whether the step is accepted.  S = T.out[0..10] and st = T.state[0..TRS_COUNT) as the
// caller fetched them at its entry (nothing writes them in between: their round trip runs under the cost's loads)
__device__ inline bool tr_decide(const TrDecideArgs& T, double cost_trial, double hl_trial, const double (&S)[11],
                                 const double (&st)[TRS_COUNT]) {
  const double gg = S[0], gHg = S[1], ww = S[2], gw = S[3], gHw = S[4], wHw = S[5], qq = S[6], hh = S[7];
  const double a = st[TRS_A], b = st[TRS_B], Delta = st[TRS_DELTA], cost = st[TRS_COST];
  int flags = (int)st[TRS_FLAGS];
  const int k = (int)st[TRS_ITER];
  const double gdqs = S[10];
  if (!__builtin_isfinite(S[9])) flags |= TRF_NONFINITE;
  const double dL_dq = gdqs / cost;   // :2517-2524
  // CalcTrustRatio (:2004-2034)
  const double gradient_term = a * gg + b * gw;
  const double hessian_term = 0.5 * (a * a * gHg + 2 * a * b * gHw + b * b * wHw);
  const double merit_k = (T.nu > 0) ? cost + S[8] : cost, merit_kp = (T.nu > 0) ? cost_trial + hl_trial : cost_trial;
  const double predicted = -gradient_term - hessian_term, actual = merit_k - merit_kp;
  const double rho = (predicted < T.eps && actual < T.eps) ? 0.5 : actual / predicted;
  if (!(dL_dq < 2.220446049250313e-16)) flags |= TRF_NOT_DESCENT;
  const bool accept = (flags == 0) && rho > T.eta;
  double* R = T.rows + (size_t)k * TRR_COUNT;
  R[TRR_COST] = cost; R[TRR_DELTA] = Delta; R[TRR_RHO] = rho; R[TRR_QNORM] = __builtin_sqrt(qq);
  R[TRR_DQNORM] = __builtin_sqrt(S[9]); R[TRR_DQHNORM] = __builtin_sqrt(ww); R[TRR_GNORM] = __builtin_sqrt(gg);
  R[TRR_DLDQ] = dL_dq; R[TRR_HNORM] = __builtin_sqrt(hh); R[TRR_ACCEPTED] = accept ? 1.0 : 0.0;
  R[TRR_CLOCK] = (double)wall_clock64(); R[TRR_A] = a; R[TRR_B] = b; R[TRR_COST_TRIAL] = cost_trial;
  R[TRR_FLAGS] = (double)flags;
  R[TRR_MERIT] = merit_k;
  R[TRR_REASON] = 0.0;
  if (flags == 0) {
    if (accept) {   // (the convergence criteria of this step: TrConvergence)
      T.state[TRS_PREVCOST] = cost; T.state[TRS_CHECK] = 1.0; T.state[TRS_DQN] = __builtin_sqrt(S[9]);
    }
    if (accept) T.state[TRS_COST] = cost_trial;   // :2550-2553
    if (rho < 0.25) T.state[TRS_DELTA] = Delta * 0.25;                                                        // :2614-2617
    else if (rho > 0.75 && st[TRS_ACTIVE] != 0.0) T.state[TRS_DELTA] = __builtin_fmin(2 * Delta, T.Delta_max);   // :2618-2622
  }
  T.state[TRS_FLAGS] = (double)flags;
  T.state[TRS_ITER] = (double)(k + 1);
  T.state[TRS_ACCEPTED] = accept ? 1.0 : 0.0;
  if (accept) T.state[TRS_CUR] = (st[TRS_CUR] != 0.0) ? 0.0 : 1.0;   // the trial point's set is the iterate's now
  return accept;
}

// The start of idto_hip_tr_solve for one problem: the loop's state words [Delta0, the cost of the resident q, accepted = 1,
// zeros] and the iteration kernel's arrival counter in ONE launch, Delta0 as a kernel argument (round 4: a fill, a copy
// from pinned words and a 2-D copy of the cost - three serialised device operations of ~4.5 us each at the head of
// every MPC re-plan).
__global__ void tr_begin_kernel(double* state, int nstate, int i_delta, int i_accepted, int i_cost, double Delta0,
                                const double* cost, unsigned long long* cnt) {
  const int i = threadIdx.x;
  if (i < nstate) state[i] = i == i_delta ? Delta0 : i == i_accepted ? 1.0 : i == i_cost ? *cost : 0.0;
  if (i == 0) *cnt = 0ull;
}

// The end of idto_hip_tr_solve for a caller that wants the solution at once (idto_hip_tr_solve_fetch): the loop's state
// words, its statistics rows and the iterate's q, v, tau (rows of the slab), the last step dq and w = H^-1 (g + J^T lambda)
// go into ONE contiguous staging buffer - v and tau from the set the iterate ended up in - so that a single copy and the
// solve's own wait bring everything to the host (round 4: seven copies, ~4 us each on the device, and a second wait: 65 us
// of an MPC re-plan).  Layout: [state TRS_COUNT | rows nrows | q | v | tau | dq | w].
struct TrGatherArgs {
  const double* state; const double* rows; int nstate, nrows;
  const double* q; const double* v; const double* slab; const double* dq; const double* w;
  int N, nq, nv, slab_stride;
  long long alt_off;   // bytes between the two output sets (0: one set)
  double* out;
};
__global__ void tr_gather_kernel(TrGatherArgs A) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
  const long long off = (A.alt_off != 0 && A.state[IDTO_TRS_CUR] != 0.0) ? A.alt_off : 0;
  const double* v = at_problem(A.v, (size_t)off);
  const double* slab = at_problem(A.slab, (size_t)off);
  const int nqa = (A.N + 1) * A.nq, nva = (A.N + 1) * A.nv, nta = A.N * A.nv;
  double* o = A.out;
  for (int i = tid; i < A.nstate; i += nt) o[i] = A.state[i];
  o += A.nstate;
  for (int i = tid; i < A.nrows; i += nt) o[i] = A.rows[i];
  o += A.nrows;
  for (int i = tid; i < nqa; i += nt) o[i] = A.q[i];
  o += nqa;
  for (int i = tid; i < nva; i += nt) o[i] = v[i];
  o += nva;
  for (int i = tid; i < nta; i += nt) { const int t = i / A.nv, r = i - t * A.nv; o[i] = slab[(size_t)t * A.slab_stride + 3 * A.nv * A.nq + r]; }
  o += nta;
  for (int i = tid; i < nqa; i += nt) o[i] = A.dq[i];
  o += nqa;
  for (int i = tid; i < nqa; i += nt) o[i] = A.w[i];
}

// batch contexts after idto_hip_tr_solve_batch: a problem whose iterate's v, a, N+, slab and products ended up in the
// second set of fd_kernel outputs (batch.h AltSel) gets them copied into the first (grid (x, problem))
__global__ void tr_fold_sets_kernel(double* set_a, size_t count, const double* state, size_t pstride) {
  const size_t o = (size_t)blockIdx.y * pstride;
  if (at_problem(state, o)[TRS_CUR] == 0.0) return;
  double* a = at_problem(set_a, o);
  const double* b = a + count;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) a[i] = b[i];
}

}  // namespace idto_dev
