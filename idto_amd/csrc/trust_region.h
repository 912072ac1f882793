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
  double* partial;
  const double* freeze;   // (resident loop) the sticky flags word: once it is non-zero D, g~, w keep the values of the last decided iteration
};

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
      if (j == 2) {
        const double wi = yi / d;
        dl[r] = d; gl[r] = gti; wl[r] = wi;
        if (!frozen) { D[v] = d; gt[v] = gti; w[v] = wi; }
      }
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
      // (eight loads in flight at a time: the rolled loop with one load per step was a chain of L2 round trips;
      // the sums keep their order)
      for (int c0 = 0; c0 < K; c0 += 8) {
        double m[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) m[u] = (c0 + u < K) ? M[r * sr + (c0 + u) * sc] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (c0 + u < K) { at += m[u] * a[c0 + u]; ay += m[u] * y[c0 + u]; }
      }
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

__global__ void __launch_bounds__(256) tr_prepare_rows_kernel(TrRowsArgs A) {
  extern __shared__ double lds[];
  tr_prepare_rows_body(A, lds);
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
  TrRowsArgs rows;
  AltSel alt;                    // rows.slab is set A's: h = tau[unactuated] is read from the iterate's set
  unsigned long long* counter;   // monotonic: workgroups of tr_iter_kernel that have published their partial sums
  unsigned long long target;     // ... its value once every workgroup of THIS launch has
  double* out;                   // [11] the nine inner products, then dq.dq and g~.D^-1 dq
  double* state;                 // [TRS_COUNT]
  int n, nq, scaling, nquat;
  const int* quat;
  double* q_trial;
  double* dq;
  TrConvergence conv;
  const unsigned* fact_status;   // the solver's status word (host-mapped) and the id of the factorisation this iteration's step
  unsigned fact_id;              // came from: a failure in the MIDDLE of the resident loop is flagged in its own iteration
  const unsigned* timeout_status;   // ... and the word a launch whose waits between workgroups ran out writes its id to
  size_t pstride, rows_stride;   // batch contexts: grid.y = problem (arena stride in bytes; doubles between the problems' rows)
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

__global__ void __launch_bounds__(256) tr_iter_kernel(TrIterArgs T) {
  extern __shared__ double lds[];
  __shared__ int last;
  __shared__ double ab[3];
  {   // problem of the batch: every array of the loop lives in the problem's arena
    const size_t o = (size_t)blockIdx.y * T.pstride;
    TrRowsArgs& A = T.rows;
    A.HA = at_problem(A.HA, o); A.HB = at_problem(A.HB, o); A.HC = at_problem(A.HC, o); A.g = at_problem(A.g, o);
    if (A.jtl) A.jtl = at_problem(A.jtl, o);
    A.yin = at_problem(A.yin, o); A.q = at_problem(A.q, o); A.Dprev = at_problem(A.Dprev, o); A.D = at_problem(A.D, o);
    A.gt = at_problem(A.gt, o); A.w = at_problem(A.w, o); A.partial = at_problem(A.partial, o);
    if (A.lambda) A.lambda = at_problem(A.lambda, o);
    if (A.freeze) A.freeze = at_problem(A.freeze, o);
    A.slab = at_problem(A.slab, o + (size_t)alt_offset(T.alt, o));   // (h = tau[unactuated] is read from the iterate's set)
    T.counter = at_problem(T.counter, o); T.out = at_problem(T.out, o); T.state = at_problem(T.state, o);
    T.q_trial = at_problem(T.q_trial, o); T.dq = at_problem(T.dq, o);
    T.conv.rows += (size_t)blockIdx.y * T.rows_stride;
    if (T.fact_status) T.fact_status += 2 * blockIdx.y;
  }
  tr_prepare_rows_body(T.rows, lds);
  const int tid = threadIdx.x, nt = blockDim.x, nblk = T.rows.nblk, n = T.n;
  if (tid == 0) {   // (block_sums left a barrier behind the partial sums of thread 0)
    __atomic_thread_fence(__ATOMIC_RELEASE);   // agent scope: the partial sums, D, g~, w of this block row
    const unsigned long long prev =
        __hip_atomic_fetch_add(T.counter, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = (prev + 1 == T.target);
  }
  __syncthreads();
  if (!last) return;
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  // (requested now, used by thread 0 at the dogleg: the round trip to host-mapped memory overlaps the sums)
  const unsigned fact_word = (tid == 0 && T.fact_status) ? __hip_atomic_load(T.fact_status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0u;
  const unsigned timeout_word = (tid == 0 && T.timeout_status) ? __hip_atomic_load(T.timeout_status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0u;
  // the operands of the trial point (below) do not depend on the dogleg: fetch them now, four passes of 256
  // threads = tr_trial_kernel's 1024, so that their L2 round trips overlap the sums and the dogleg instead of
  // following them one pass after the other
  constexpr int NPASS = 4;
  double pg[NPASS], pw[NPASS], pd[NPASS], pq[NPASS];
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    const int idx = tid + 256 * ps;
    pg[ps] = pw[ps] = pd[ps] = pq[ps] = 0.0;
    if (idx < n) {
      pg[ps] = T.rows.gt[idx]; pw[ps] = T.rows.w[idx]; pq[ps] = T.rows.q[idx];
      if (T.scaling) pd[ps] = T.rows.D[idx];
    }
  }
  // ---- tr_prepare_sum: the partial sums in block order
  double* part = lds;            // [9 nblk]
  double* S = part + 9 * nblk;   // [9]
  double* scratch = S + 9;       // [2 * 16]
  for (int idx = tid; idx < 9 * nblk; idx += nt) part[idx] = T.rows.partial[idx];
  for (int idx = tid; idx < n; idx += nt) const_cast<double*>(T.rows.Dprev)[idx] = T.rows.D[idx];   // the adaptive methods' memory
  __syncthreads();
  if (tid < 9) {
    double acc = 0.0;
    for (int i = 0; i < nblk; ++i) acc += part[i * 9 + tid];
    S[tid] = acc;
    T.out[tid] = acc;
  }
  __syncthreads();
  // ---- the convergence criteria of the step the previous iteration accepted (TrConvergence): g.dq with the merit
  // function's gradient at THIS iterate and the dq that led here (still in T.dq: the trial point below overwrites it)
  if (T.conv.on) {
    double sgd = 0.0;
    for (int idx = tid; idx < n; idx += nt) {
      const double gm = T.rows.jtl ? T.rows.g[idx] + T.rows.jtl[idx] : T.rows.g[idx];
      sgd += gm * T.dq[idx];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sgd += __shfl_down(sgd, off);
    if ((tid & 63) == 0) scratch[tid >> 6] = sgd;
    __syncthreads();
    if (tid == 0) {
      int flags = (int)T.state[TRS_FLAGS];
      if (T.state[TRS_CHECK] != 0.0 && (flags & ~TRF_CONVERGED) == 0) {
        double gdq = 0.0;
        for (int wv = 0; wv < nt / 64; ++wv) gdq += scratch[wv];
        const double cost = T.state[TRS_COST], prev = T.state[TRS_PREVCOST];
        int reason = 0;
        if (__builtin_fabs(prev - cost) < T.conv.abs_cost + T.conv.rel_cost * cost) reason |= 1;
        if (__builtin_fabs(gdq) < T.conv.abs_grad + T.conv.rel_grad * cost) reason |= 2;
        if (T.state[TRS_DQN] < T.conv.abs_state + T.conv.rel_state * __builtin_sqrt(S[6])) reason |= 4;
        const int k_prev = (int)T.state[TRS_ITER] - 1;
        if (k_prev >= 0) T.conv.rows[(size_t)k_prev * TRR_COUNT + TRR_REASON] = (double)reason;
        if (reason) T.state[TRS_FLAGS] = (double)(flags | TRF_CONVERGED);
      }
      T.state[TRS_CHECK] = 0.0;
    }
    __syncthreads();
    if (T.conv.check_only) return;
  }
  // ---- CalcDoglegPoint normalised by Delta: pU = cU g~ (TO.cc:2157), pH = -w / Delta (:2139-2149)
  if (tid == 0) {
    const double gg = S[0], gHg = S[1], ww = S[2], gw = S[3];
    const double Delta = T.state[TRS_DELTA];
    int flags = (int)T.state[TRS_FLAGS];
    if (T.fact_status && fact_word == T.fact_id) flags |= TRF_FACTORIZATION;
    if (T.timeout_status && timeout_word == T.fact_id) flags |= TRF_SOLVER_TIMEOUT;
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
    T.state[TRS_A] = a; T.state[TRS_B] = b; T.state[TRS_ACTIVE] = active; T.state[TRS_FLAGS] = (double)flags;
    ab[0] = a; ab[1] = b; ab[2] = (double)flags;
  }
  __syncthreads();
  // ---- tr_trial: dq = D (a g~ + b w), q_trial = q + dq, [dq.dq, g~.(a g~ + b w)] with the partial sums
  // of tr_trial_kernel's 1024 threads (thread v of it owns idx = v, v + 1024, ...): same bits
  // (once a sticky flag is set - converged, or an error - the loop idles: dq and the trial point keep the values of the
  // last decided iteration, which is what the warm start hands on: TO.cc:2361-2385)
  const double a = ab[0], b = ab[1];
  const bool idle = ab[2] != 0.0;
  const int lane = tid & 63;
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {   // (blockDim.x == 256)
    const int vt = tid + 256 * ps;
    double s0 = 0.0, s1 = 0.0;
    if (vt < n) {   // the prefetched first element of virtual thread vt
      const double dqs = a * pg[ps] + b * pw[ps];
      const double dq = T.scaling ? pd[ps] * dqs : dqs;
      if (!idle) { T.dq[vt] = dq; T.q_trial[vt] = pq[ps] + dq; }
      s0 += dq * dq;
      s1 += pg[ps] * dqs;
    }
    for (int idx = vt + 1024; idx < n; idx += 1024) {
      const double dqs = a * T.rows.gt[idx] + b * T.rows.w[idx];
      const double dq = T.scaling ? T.rows.D[idx] * dqs : dqs;
      if (!idle) { T.dq[idx] = dq; T.q_trial[idx] = T.rows.q[idx] + dq; }
      s0 += dq * dq;
      s1 += T.rows.gt[idx] * dqs;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { s0 += __shfl_down(s0, off); s1 += __shfl_down(s1, off); }
    if (lane == 0) { scratch[vt >> 6] = s0; scratch[16 + (vt >> 6)] = s1; }
  }
  __syncthreads();
  if (T.nquat > 0 && !idle) {
    const int nsteps = n / T.nq;
    for (int idx = tid; idx < nsteps * T.nquat; idx += nt) {
      const int t = idx / T.nquat, qs = T.quat[idx - t * T.nquat];
      double* qq = T.q_trial + (size_t)t * T.nq + qs;
      const double nrm = __builtin_sqrt(qq[0] * qq[0] + qq[1] * qq[1] + qq[2] * qq[2] + qq[3] * qq[3]);
      for (int k = 0; k < 4; ++k) qq[k] /= nrm;
    }
  }
  if (tid == 0) {
    double x0 = 0.0, x1 = 0.0;
    for (int wv = 0; wv < 16; ++wv) { x0 += scratch[wv]; x1 += scratch[16 + wv]; }
    T.out[9] = x0; T.out[10] = x1;
  }
}

struct TrDecideArgs {
  double* state;        // [TRS_COUNT]
  const double* out;    // [11] from tr_iter_kernel
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
// constraints are enforced); returns whether the step is accepted
__device__ inline bool tr_decide(const TrDecideArgs& T, double cost_trial, double hl_trial) {
  const double* S = T.out;
  const double gg = S[0], gHg = S[1], ww = S[2], gw = S[3], gHw = S[4], wHw = S[5], qq = S[6], hh = S[7];
  const double a = T.state[TRS_A], b = T.state[TRS_B], Delta = T.state[TRS_DELTA], cost = T.state[TRS_COST];
  int flags = (int)T.state[TRS_FLAGS];
  const int k = (int)T.state[TRS_ITER];
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
    else if (rho > 0.75 && T.state[TRS_ACTIVE] != 0.0) T.state[TRS_DELTA] = __builtin_fmin(2 * Delta, T.Delta_max);   // :2618-2622
  }
  T.state[TRS_FLAGS] = (double)flags;
  T.state[TRS_ITER] = (double)(k + 1);
  T.state[TRS_ACCEPTED] = accept ? 1.0 : 0.0;
  if (accept) T.state[TRS_CUR] = (T.state[TRS_CUR] != 0.0) ? 0.0 : 1.0;   // the trial point's set is the iterate's now
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
