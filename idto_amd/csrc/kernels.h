// kernels.h — the three kernels of one Gauss-Newton iteration on gfx950:
//   fd_kernel        N+ / v / a / tau and the finite-difference partials  (a1-a9 of SURVEY.md §8a)
//   assemble_kernel  gradient and Gauss-Newton Hessian bands              (a11, a12)
//   penta_kernel     block-Thomas factorisation + solve                   (a14, a15)
// plus small helpers (cost, multi-RHS solve).  fp64 throughout; compiled with
// -ffp-contract=off; every sum has the association order of DESIGN.md §3.2.
#pragma once

#include "fd_kernel.h"
#include "trust_region.h"

namespace idto_dev {

// ---------------------------------------------------------------------------
// Cost L(q) from resident q, v, tau (TO.cc:147-176).  One block per problem; the final sum
// is accumulated serially in the reference's order.
// cost_body: the workgroup's work, every pointer the problem's already (cost_kernel below; one more workgroup of the
// pipelined solver's launch inside the trust-region loop, penta_pipe.h PipeDecide).  Returns - with T.state set - whether
// the step was accepted (the same value in every thread).
// TU: time steps a thread takes per pass of the columns (32 x TU x blockDim.x / 1024 steps a pass: one pass = one round trip)
template <int TU = 2>
__device__ __forceinline__ bool cost_body(const int nq_, const int nv_, const DevProblem& P, const double* __restrict__ q,
                                          const double* __restrict__ v, const double* __restrict__ slab, int slab_stride,
                                          double* __restrict__ cost_out, int diag, double* __restrict__ pack,
                                          double* __restrict__ cost_copy, const TrDecideArgs& T) {
  // e^T W e per term as the reference's Eigen expression evaluates it: tot = sum_c (sum_r e_r W[r][c]) e_c.
  // One thread per (term, column c); `diag`: the weights are diagonal, the inner sum is its one
  // non-zero product (the others are exact zeros).  Then one thread per term adds the columns in
  // order, and thread 0 the terms in order (TO.cc:147-176).
  extern __shared__ double lds[];
  TR_STAMP(16);
  // (trust-region loop) what the decision and an accepted step's q <- q_trial need and nothing in this launch writes:
  // requested here, their round trips run under the cost's own
  constexpr int QPF = 2;
  double dS[11], dst[TRS_COUNT], qpf[QPF], p2v = 0.0;
  if (T.state) {
    dS[9] = dS[10] = 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) dS[k] = (threadIdx.x == 0) ? T.out[k] : 0.0;
    // (dq.dq and g~.D^-1 dq come per block row from tr_iter_kernel's workgroups: added up below, in block order)
    if ((int)threadIdx.x < 2 * T.nblk) p2v = T.part2[threadIdx.x];
#pragma unroll
    for (int k = 0; k < TRS_COUNT; ++k) dst[k] = (threadIdx.x == 0) ? T.state[k] : 0.0;
#pragma unroll
    for (int u = 0; u < QPF; ++u) { const int idx = threadIdx.x + u * blockDim.x; qpf[u] = (idx < T.n) ? T.q_trial[idx] : 0.0; }
  }
  const int N = P.N, nq = nq_, nv = nv_, tid = threadIdx.x, nt = blockDim.x;
  const int nterms = 3 * N + 2, nmax = nq > nv ? nq : nv;
  double* terms = lds;               // [nterms]
  double* cols = terms + nterms;     // [nterms][nmax]
  double* p2l = cols + nterms * nmax;   // [2 (N + 1) + 2] (trust-region loop)
  if (diag && nmax <= 32 && (nt & 31) == 0) {
    // Diagonal weights, the production case.  Thread (c = tid & 31, tl = tid >> 5) takes column c of the (up to) three
    // terms of the steps t = tl, tl + nt / 32, ...: the loads of a step are consecutive in c, every operand's address is
    // a multiply-add of (t, c) - no division, no branch per kind - and the loads of a pass are all in flight before the
    // first is used.  (One thread per (term, column) item of a flat index spent 800 instructions a wavefront on
    // idx / nmax, term % 3 and five-way pointer selects: 16 wavefronts on one CU, the phase was issue-bound at 4.4 us of
    // this kernel's 7.5.)  Same expressions per item as below, same bits.
    const int c = tid & 31, tl = tid >> 5, tstep = nt >> 5;
    const bool cq = c < nq, cv = c < nv;
    const int cqi = cq ? c : 0, cvi = cv ? c : 0;
    // (the weights of this thread's column: five loads, once)
    const double wq = P.Qq0[cqi * nq + cqi], wv = P.Qv0[cvi * nv + cvi], wr = P.R0[cvi * nv + cvi];
    const double wfq = P.Qfq0[cqi * nq + cqi], wfv = P.Qfv0[cvi * nv + cvi];
    const double* tau0 = slab + 3 * nv * nq;
    for (int t0 = tl; t0 <= N; t0 += TU * tstep) {
      double eq[TU], nq_[TU], ev[TU], nv_[TU], et[TU];
#pragma unroll
      for (int u = 0; u < TU; ++u) {
        const int t = t0 + u * tstep;
        const bool ok = t <= N, run = t < N;
        const int tt = ok ? t : 0;
        eq[u] = q[tt * nq + cqi]; nq_[u] = P.q_nom[tt * nq + cqi];
        ev[u] = v[tt * nv + cvi]; nv_[u] = P.v_nom[tt * nv + cvi];
        et[u] = tau0[(size_t)(run ? t : 0) * slab_stride + cvi];
      }
#pragma unroll
      for (int u = 0; u < TU; ++u) {
        const int t = t0 + u * tstep;
        const bool ok = t <= N, run = t < N;
        double valq, valv, valt;
        { const double dc = eq[u] - nq_[u]; double acc = 0; acc += dc * (run ? wq : wfq); valq = cq ? acc * dc : 0.0; }
        { const double dc = ev[u] - nv_[u]; double acc = 0; acc += dc * (run ? wv : wfv); valv = cv ? acc * dc : 0.0; }
        { const double dc = et[u] - 0.0; double acc = 0; acc += dc * wr; valt = cv ? acc * dc : 0.0; }
        if (ok && c < nmax) {
          const int term = run ? 3 * t : 3 * N;
          cols[term * nmax + c] = valq;
          cols[(term + 1) * nmax + c] = valv;
          if (run) cols[(term + 2) * nmax + c] = valt;
        }
      }
    }
  } else if (diag) {
    // (blocks wider than 32 - none of the examples -: one thread per (term, column) item; the three operands of an item -
    // value, nominal value, weight - are independent loads, four items' worth are issued before the first is waited for)
    constexpr int U = 4;
    for (int idx0 = tid; idx0 < nterms * nmax; idx0 += U * nt) {
      double ev[U], en[U], wv[U];
      bool use[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = idx0 + u * nt;
        const bool ok = idx < nterms * nmax;
        const int term = ok ? idx / nmax : 0, c = ok ? idx - term * nmax : 0;
        const int kind = (term < 3 * N) ? term % 3 : 3 + (term - 3 * N);
        const int t = (term < 3 * N) ? term / 3 : N;
        const int n = (kind == 0 || kind == 3) ? nq : nv;
        use[u] = ok && c < n;
        const int cc = use[u] ? c : 0;
        const double* pe = (kind == 0 || kind == 3) ? q + t * nq + cc : (kind == 1 || kind == 4) ? v + t * nv + cc
                                                                     : slab + (size_t)t * slab_stride + 3 * nv * nq + cc;
        const double* pn = (kind == 0 || kind == 3) ? P.q_nom + t * nq + cc : (kind == 1 || kind == 4) ? P.v_nom + t * nv + cc : pe;
        const double* W = (kind == 0) ? P.Qq0 : (kind == 1) ? P.Qv0 : (kind == 2) ? P.R0 : (kind == 3) ? P.Qfq0 : P.Qfv0;
        ev[u] = *pe; en[u] = *pn; wv[u] = W[cc * n + cc];
        en[u] = (kind == 2) ? 0.0 : en[u];   // (tau's nominal value is zero: e = tau - 0.0)
      }
      // (the loads stay where they are: without the pin the compiler sinks them into the `use` branches below, a wait each)
      asm volatile("" : "+v"(ev[0]), "+v"(ev[1]), "+v"(ev[2]), "+v"(ev[3]), "+v"(wv[0]), "+v"(wv[1]), "+v"(wv[2]), "+v"(wv[3]));
      asm volatile("" : "+v"(en[0]), "+v"(en[1]), "+v"(en[2]), "+v"(en[3]));
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = idx0 + u * nt;
        if (idx >= nterms * nmax) continue;
        double val = 0.0;
        if (use[u]) {
          const double dc = ev[u] - en[u];
          double acc = 0;
          acc += dc * wv[u];
          val = acc * dc;
        }
        cols[idx] = val;
      }
    }
  } else
  for (int idx = tid; idx < nterms * nmax; idx += nt) {
    const int term = idx / nmax, c = idx - term * nmax;
    // kind of the term: 0 q_t, 1 v_t, 2 tau_t (running), 3 q_N, 4 v_N (terminal)
    const int kind = (term < 3 * N) ? term % 3 : 3 + (term - 3 * N);
    const int t = (term < 3 * N) ? term / 3 : N;
    const int n = (kind == 0 || kind == 3) ? nq : nv;
    auto err = [&](int r) {  // e_r - nominal_r
      if (kind == 0 || kind == 3) return q[t * nq + r] - P.q_nom[t * nq + r];
      if (kind == 1 || kind == 4) return v[t * nv + r] - P.v_nom[t * nv + r];
      return slab[(size_t)t * slab_stride + 3 * nv * nq + r] - 0.0;
    };
    const double* W = (kind == 0) ? P.Qq0 : (kind == 1) ? P.Qv0 : (kind == 2) ? P.R0 : (kind == 3) ? P.Qfq0 : P.Qfv0;
    double val = 0.0;
    if (c < n) {
      const double dc = err(c);
      double acc = 0;
      for (int r = 0; r < n; ++r) acc += err(r) * W[c * n + r];
      val = acc * dc;
    }
    cols[idx] = val;
  }
  if (T.state) {
    if (tid < 2 * T.nblk) p2l[tid] = p2v;
    for (int idx = tid + nt; idx < 2 * T.nblk; idx += nt) p2l[idx] = T.part2[idx];
  }
  __syncthreads();
  TR_STAMP(17);
  for (int term = tid; term < nterms; term += nt) {
    const int n = (term < 3 * N) ? ((term % 3 == 0) ? nq : nv) : ((term == 3 * N) ? nq : nv);
    double tot = 0;
    for (int c = 0; c < n; ++c) tot += cols[term * nmax + c];
    terms[term] = tot;
  }
  __syncthreads();
  TR_STAMP(18);
  if (T.state && (tid == 64 || tid == 128)) {   // (beside thread 0's chain of adds)
    const int k = tid == 64 ? 0 : 1;
    double acc = 0.0;
    for (int b = 0; b < T.nblk; ++b) acc += p2l[2 * b + k];
    p2l[2 * T.nblk + k] = acc;
  }
  double cost = 0;
  if (tid == 0) {
    int i = 0;
    for (; i + 16 <= 3 * N; i += 16) {   // the terms in order, the LDS reads sixteen at a time ahead of the chain of adds
      double t16[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) t16[u] = terms[i + u];
#pragma unroll
      for (int u = 0; u < 16; ++u) cost += t16[u];
    }
    for (; i + 8 <= 3 * N; i += 8) {
      double t8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t8[u] = terms[i + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) cost += t8[u];
    }
    for (; i < 3 * N; ++i) cost += terms[i];
    cost *= P.dt;
    cost += terms[3 * N];
    cost += terms[3 * N + 1];
    *cost_out = cost;
    if (cost_copy) *cost_copy = cost;   // (single-problem contexts: next to the other trust-region scalars)
    if (pack) pack[N * nv] = cost;
  }
  TR_STAMP(19);
  if (T.state) {   // (idto_hip_tr_solve) this was the trial point of a trust-region iteration
    const int neq = T.nu * T.N;
    __syncthreads();
    if (tid == 0) { dS[9] = p2l[2 * T.nblk]; dS[10] = p2l[2 * T.nblk + 1]; T.out[9] = dS[9]; T.out[10] = dS[10]; }
    if (T.nu > 0) {   // h(q + dq) . lambda: products by everybody, added in index order by thread 0
      __syncthreads();
      for (int r = tid; r < neq; r += nt) {
        const int t = r / T.nu, j = r - t * T.nu;
        cols[r] = slab[(size_t)t * slab_stride + T.tau_off + T.dofs[j]] * T.lambda[r];
      }
      __syncthreads();
    }
    if (tid == 0) {   // ratio, accept / reject, radius
      double hl = 0.0;
      if (T.nu > 0) {   // (in index order, the LDS reads eight at a time ahead of the chain of adds)
        int r = 0;
        for (; r + 8 <= neq; r += 8) {
          double c8[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) c8[u] = cols[r + u];
#pragma unroll
          for (int u = 0; u < 8; ++u) hl += c8[u];
        }
        for (; r < neq; ++r) hl += cols[r];
      }
      terms[0] = tr_decide(T, cost, hl, dS, dst) ? 1.0 : 0.0;
    }
    TR_STAMP(20);
  }
  bool accepted = false;
  if (T.state) {
    __syncthreads();
    accepted = terms[0] != 0.0;
    if (accepted) {
#pragma unroll
      for (int u = 0; u < QPF; ++u) { const int idx = tid + u * nt; if (idx < T.n) T.q[idx] = qpf[u]; }
      for (int idx = tid + QPF * nt; idx < T.n; idx += nt) T.q[idx] = T.q_trial[idx];
    }
    TR_STAMP(21);
  }
  // [tau_0 .. tau_{N-1} | cost] contiguous: what a trial point of the trust-region loop reads back
  if (pack)
    for (int idx = tid; idx < N * nv; idx += nt) {
      const int t = idx / nv, r = idx - t * nv;
      pack[idx] = slab[(size_t)t * slab_stride + 3 * nv * nq + r];
    }
  return accepted;
}

__global__ void cost_kernel(DevModel M, DevProblem P, const double* __restrict__ q, const double* __restrict__ v,
                            const double* __restrict__ slab, int slab_stride, double* __restrict__ cost_out,
                            int diag, double* __restrict__ pack, size_t pstride, double* __restrict__ cost_copy,
                            TrDecideArgs T, AltSel alt) {
  {
    const size_t o = (size_t)blockIdx.y * pstride, w = o + (size_t)alt_offset(alt, o);
    P = at_problem(P, o); q = at_problem(q, o); v = at_problem(v, w); slab = at_problem(slab, w);
    cost_out = at_problem(cost_out, o);
    if (pack) pack = at_problem(pack, o);
    if (T.state) {   // (the trust-region decision of THIS problem: trust_region.h tr_decide)
      T.state = at_problem(T.state, o); T.out = at_problem(T.out, o); T.q = at_problem(T.q, o);
      T.q_trial = at_problem(T.q_trial, o); T.rows += (size_t)blockIdx.y * T.rows_stride;
      T.part2 = at_problem(T.part2, o);
      if (T.lambda) T.lambda = at_problem(T.lambda, o);
    }
  }
  (void)cost_body(M.nq, M.nv, P, q, v, slab, slab_stride, cost_out, diag, pack, cost_copy, T);
}

// ---------------------------------------------------------------------------
// assemble_kernel: block <-> block row i of the Hessian (and g_i).
// Restates TO.cc:1021-1081 and :1093-1165 row by row (see DESIGN.md §4.2 for
// the re-indexing from the reference's column-wise loop).
IDTO_DEV void acc_atwb(const double* A, const double* W, const double* B, double* AtW, double* out, bool init,
                       bool lower_only, int nq, int nv, int tid, int nt) {
  // AtW = A^T W  (nq x nv), then out (+)= AtW B
  for (int idx = tid; idx < nq * nv; idx += nt) {
    const int c = idx / nq, r = idx - c * nq;
    double acc = A[r * nv] * W[c * nv];
    for (int l = 1; l < nv; ++l) acc += A[r * nv + l] * W[c * nv + l];
    AtW[c * nq + r] = acc;
  }
  __syncthreads();
  for (int idx = tid; idx < nq * nq; idx += nt) {
    const int c = idx / nq, r = idx - c * nq;
    if (lower_only && r < c) continue;
    double acc = AtW[r] * B[c * nv];
    for (int l = 1; l < nv; ++l) acc += AtW[l * nq + r] * B[c * nv + l];
    out[c * nq + r] = init ? acc : out[c * nq + r] + acc;
  }
  __syncthreads();
}

IDTO_DEV void acc_vec_w_mat(const double* e, const double* W, const double* J, double* tmp, double* out, bool init,
                            int nq, int n, int tid, int nt) {
  for (int r = tid; r < n; r += nt) {
    double acc = e[0] * W[r * n];
    for (int i = 1; i < n; ++i) acc += e[i] * W[r * n + i];
    tmp[r] = acc;
  }
  __syncthreads();
  for (int j = tid; j < nq; j += nt) {
    double acc = tmp[0] * J[j * n];
    for (int r = 1; r < n; ++r) acc += tmp[r] * J[j * n + r];
    out[j] = init ? acc : out[j] + acc;
  }
  __syncthreads();
}

__global__ void assemble_kernel(DevModel M, DevProblem P, const double* __restrict__ q,
                                const double* __restrict__ slab, int slab_stride, double* __restrict__ g,
                                double* __restrict__ HA, double* __restrict__ HB, double* __restrict__ HC,
                                size_t pstride) {
  {
    const size_t o = (size_t)blockIdx.y * pstride;
    P = at_problem(P, o); q = at_problem(q, o); slab = at_problem(slab, o); g = at_problem(g, o);
    HA = at_problem(HA, o); HB = at_problem(HB, o); HC = at_problem(HC, o);
  }
  extern __shared__ double lds[];
  const int tid = threadIdx.x, nt = blockDim.x;
  const int i = blockIdx.x, N = P.N, nq = M.nq, nv = M.nv;
  const int bsz = nv * nq, qq = nq * nq;
  const double dt = P.dt;

  double* qm1 = lds;
  double* q0 = qm1 + nq;
  double* q1 = q0 + nq;
  double* N0 = q1 + nq;       // N+_i
  double* N1 = N0 + bsz;      // N+_{i+1}
  double* Vi = N1 + bsz;      // dv_i/dq_i
  double* Wi = Vi + bsz;      // dv_i/dq_{i-1}
  double* Wi1 = Wi + bsz;     // dv_{i+1}/dq_i
  double* v0 = Wi1 + bsz;     // v_i
  double* v1 = v0 + nv;       // v_{i+1}
  double* ve = v1 + nv;       // v_i - vnom_i
  double* vep = ve + nv;      // v_{i+1} - vnom_{i+1}
  double* qe = vep + nv;      // q_i - qnom_i
  double* tmp = qe + nq;      // [max(nq,nv)]
  double* AtW = tmp + (nq > nv ? nq : nv);  // nq x nv
  double* Cb = AtW + bsz;     // nq x nq
  double* Bb = Cb + qq;
  double* Ab = Bb + qq;
  double* gb = Ab + qq;       // nq

  double* Cg = HC + (size_t)i * qq;
  double* Bg = HB + (size_t)i * qq;
  double* Ag = HA + (size_t)i * qq;

  if (i == 0) {  // q_0 is fixed: C_0 = I, g_0 = 0 (TO.cc:1124, 1044)
    for (int idx = tid; idx < qq; idx += nt) {
      Cg[idx] = (idx / nq == idx % nq) ? 1.0 : 0.0;
      Bg[idx] = 0.0;
      Ag[idx] = 0.0;
    }
    for (int j = tid; j < nq; j += nt) g[j] = 0.0;
    return;
  }

  for (int c = tid; c < nq; c += nt) {
    qm1[c] = q[(i - 1) * nq + c];
    q0[c] = q[i * nq + c];
    q1[c] = (i < N) ? q[(i + 1) * nq + c] : 0.0;
  }
  __syncthreads();
  nplus_block(M, q0, N0, tid, nt);
  velocity_block(M, N0, q0, qm1, dt, v0, tid, nt);
  if (i < N) {
    nplus_block(M, q1, N1, tid, nt);
    velocity_block(M, N1, q1, q0, dt, v1, tid, nt);
  }
  __syncthreads();
  const double idt = 1 / dt, midt = -1 / dt;
  for (int idx = tid; idx < bsz; idx += nt) {
    Vi[idx] = idt * N0[idx];
    Wi[idx] = midt * N0[idx];
    Wi1[idx] = (i < N) ? midt * N1[idx] : 0.0;
  }
  for (int r = tid; r < nv; r += nt) {
    ve[r] = v0[r] - P.v_nom[i * nv + r];
    vep[r] = (i < N) ? v1[r] - P.v_nom[(i + 1) * nv + r] : 0.0;
  }
  for (int c = tid; c < nq; c += nt) qe[c] = q0[c] - P.q_nom[i * nq + c];
  __syncthreads();

  auto Mk = [&](int k) { return slab + (size_t)k * slab_stride; };
  auto Tk = [&](int k) { return slab + (size_t)k * slab_stride + bsz; };
  auto Pk = [&](int k) { return slab + (size_t)k * slab_stride + 2 * bsz; };
  auto tauk = [&](int k) { return slab + (size_t)k * slab_stride + 3 * bsz; };

  // ---- diagonal block C_i (lower triangle computed, mirrored on store)
  if (i < N) {
    for (int idx = tid; idx < qq; idx += nt) Cb[idx] = P.Qq[idx];                       // :1128
    __syncthreads();
    acc_atwb(Vi, P.Qv, Vi, AtW, Cb, false, true, nq, nv, tid, nt);                      // :1129
    acc_atwb(Pk(i - 1), P.R, Pk(i - 1), AtW, Cb, false, true, nq, nv, tid, nt);         // :1130
    acc_atwb(Tk(i), P.R, Tk(i), AtW, Cb, false, true, nq, nv, tid, nt);                 // :1131
    if (i < N - 1) {
      acc_atwb(Mk(i + 1), P.R, Mk(i + 1), AtW, Cb, false, true, nq, nv, tid, nt);       // :1133
      acc_atwb(Wi1, P.Qv, Wi1, AtW, Cb, false, true, nq, nv, tid, nt);                  // :1134
    } else {
      acc_atwb(Wi1, P.Qfv, Wi1, AtW, Cb, false, true, nq, nv, tid, nt);                 // :1136
    }
  } else {
    for (int idx = tid; idx < qq; idx += nt) Cb[idx] = P.Qfq[idx];                      // :1158
    __syncthreads();
    acc_atwb(Vi, P.Qfv, Vi, AtW, Cb, false, true, nq, nv, tid, nt);                     // :1159
    acc_atwb(Pk(N - 1), P.R, Pk(N - 1), AtW, Cb, false, true, nq, nv, tid, nt);         // :1160-1161
  }
  // ---- B_i = d g_i / d q_{i-1}   (reference index t = i-1, TO.cc:1140-1147)
  if (i >= 2) {
    acc_atwb(Pk(i - 1), P.R, Tk(i - 1), AtW, Bb, true, false, nq, nv, tid, nt);         // :1141
    if (i < N) {
      acc_atwb(Tk(i), P.R, Mk(i), AtW, Bb, false, false, nq, nv, tid, nt);              // :1143
      acc_atwb(Vi, P.Qv, Wi, AtW, Bb, false, false, nq, nv, tid, nt);                   // :1144
    } else {
      acc_atwb(Vi, P.Qfv, Wi, AtW, Bb, false, false, nq, nv, tid, nt);                  // :1146
    }
  } else {
    for (int idx = tid; idx < qq; idx += nt) Bb[idx] = 0.0;
  }
  // ---- A_i = d g_i / d q_{i-2}   (reference index t = i-2, TO.cc:1150-1153)
  if (i >= 3) {
    acc_atwb(Pk(i - 1), P.R, Mk(i - 1), AtW, Ab, true, false, nq, nv, tid, nt);         // :1152
  } else {
    for (int idx = tid; idx < qq; idx += nt) Ab[idx] = 0.0;
  }
  __syncthreads();
  for (int idx = tid; idx < qq; idx += nt) {
    const int c = idx / nq, r = idx - c * nq;
    Cg[idx] = (r >= c) ? Cb[idx] : Cb[r * nq + c];  // MakeSymmetric: upper <- lower^T (penta_diagonal_matrix.cc:71-76)
    Bg[idx] = Bb[idx];
    Ag[idx] = Ab[idx];
  }

  // ---- gradient block g_i (TO.cc:1046-1080)
  if (i < N) {
    for (int j = tid; j < nq; j += nt) {                                                // :1050
      double acc = qe[0] * P.Qq[j * nq];
      for (int c = 1; c < nq; ++c) acc += qe[c] * P.Qq[j * nq + c];
      gb[j] = acc;
    }
    __syncthreads();
    acc_vec_w_mat(ve, P.Qv, Vi, tmp, gb, false, nq, nv, tid, nt);                                   // :1053
    acc_vec_w_mat(vep, (i == N - 1) ? P.Qfv : P.Qv, Wi1, tmp, gb, false, nq, nv, tid, nt);          // :1054-1061
    acc_vec_w_mat(tauk(i - 1), P.R, Pk(i - 1), tmp, gb, false, nq, nv, tid, nt);                    // :1064
    acc_vec_w_mat(tauk(i), P.R, Tk(i), tmp, gb, false, nq, nv, tid, nt);                            // :1065
    if (i != N - 1) acc_vec_w_mat(tauk(i + 1), P.R, Mk(i + 1), tmp, gb, false, nq, nv, tid, nt);    // :1068
  } else {
    acc_vec_w_mat(tauk(N - 1), P.R, Pk(N - 1), tmp, gb, true, nq, nv, tid, nt);                     // :1075-1076
    for (int j = tid; j < nq; j += nt) {                                                            // :1077-1078
      double acc = qe[0] * P.Qfq[j * nq];
      for (int c = 1; c < nq; ++c) acc += qe[c] * P.Qfq[j * nq + c];
      gb[j] = gb[j] + acc;
    }
    __syncthreads();
    acc_vec_w_mat(ve, P.Qfv, Vi, tmp, gb, false, nq, nv, tid, nt);                                  // :1079-1080
  }
  for (int j = tid; j < nq; j += nt) g[(size_t)i * nq + j] = gb[j];
}

// ---------------------------------------------------------------------------
// assemble_diag_kernel: the same block row i as assemble_kernel for the case where all
// five weight matrices are diagonal (every example of the reference builds them with
// `.asDiagonal()`, examples/example_base.cc:387-391).  (A^T W) is then A(l, r) * w_l,
// which is what the dense product yields bit for bit (the other terms of that sum are
// exact zeros), so both kernels produce identical results.  All operands are staged in
// LDS once; each thread owns a few output elements and runs their terms in the
// reference's order without intermediate barriers.
// Grid (N + 1, 4): block row i is split over four workgroups (41 block rows alone would use 41
// of 256 CUs): part 0 the diagonal block C_i (lower triangle, mirrored), parts 1 and 2 the two
// column halves of B_i, part 3 A_i and the gradient block g_i.  Every part stages all operands
// (they are L2-resident).  The weighted operands (A(l, r) * w_l) are formed once per block in
// LDS - the same product the inline expression yields - so that the inner loops are pure
// 16-byte LDS reads + mul + add in the reference's term order.
IDTO_DEV void assemble_diag_body(const DevModel& M, const DevProblem& P, const double* __restrict__ q,
                                 const double* __restrict__ slab, int slab_stride, double* __restrict__ g,
                                 double* __restrict__ HA, double* __restrict__ HB, double* __restrict__ HC,
                                 int stop_after, const double* __restrict__ v_res,
                                 const double* __restrict__ nplus_res, const int i, const int part) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x, nt = blockDim.x;
  const int N = P.N, nq = M.nq, nv = M.nv;
  const int bsz = nv * nq, qq = nq * nq;
  const int nvp = (nv + 1) & ~1;      // padded column length: 16-byte aligned columns in LDS
  const int psz = nvp * nq;
  const double dt = P.dt;
  double* Cg = HC + (size_t)i * qq;
  double* Bg = HB + (size_t)i * qq;
  double* Ag = HA + (size_t)i * qq;
  if (i == 0) {
    if (part == 0) {
      for (int idx = tid; idx < qq; idx += nt) {
        Cg[idx] = (idx / nq == idx % nq) ? 1.0 : 0.0;
        Bg[idx] = 0.0;
        Ag[idx] = 0.0;
      }
      for (int j = tid; j < nq; j += nt) g[j] = 0.0;
    }
    return;
  }
  // operand slots (nv x nq each, column stride nvp): raw, then weighted
  enum { S_PM1 = 0, S_TM1, S_T0, S_MM1, S_M0, S_MP1, S_V, S_W, S_W1, X_V, X_P, X_T, X_M, X_W1, S_COUNT };
  enum { W_QV = 0, W_R, W_QFV, W_COUNT };
  double* ops = lds;                       // S_COUNT * psz
  double* wts = ops + S_COUNT * psz;       // W_COUNT * nv
  double* wq = wts + W_COUNT * nv;         // Qq' diag (nq)
  double* wfq = wq + nq;                   // Qfq' diag (nq)
  double* qm1 = wfq + nq;
  double* q0 = qm1 + nq;
  double* q1 = q0 + nq;
  double* N0 = q1 + nq;                    // bsz
  double* N1 = N0 + bsz;                   // bsz
  double* v0 = N1 + bsz;
  double* v1 = v0 + nv;
  double* ve = v1 + nv;
  double* vep = ve + nv;
  double* qe = vep + nv;                   // nq
  double* taus = qe + nq;                  // 3 * nv: tau_{i-1}, tau_i, tau_{i+1}

  auto blk = [&](int k, int which) { return slab + (size_t)k * slab_stride + which * bsz; };  // 0 M, 1 T, 2 P
  for (int idx = tid; idx < bsz; idx += nt) {
    const int c = idx / nv, l = idx - c * nv, o = c * nvp + l;
    ops[S_PM1 * psz + o] = blk(i - 1, 2)[idx];
    ops[S_TM1 * psz + o] = blk(i - 1, 1)[idx];
    ops[S_MM1 * psz + o] = (i >= 3) ? blk(i - 1, 0)[idx] : 0.0;
    ops[S_T0 * psz + o] = (i < N) ? blk(i, 1)[idx] : 0.0;
    ops[S_M0 * psz + o] = (i < N && i >= 2) ? blk(i, 0)[idx] : 0.0;
    ops[S_MP1 * psz + o] = (i < N - 1) ? blk(i + 1, 0)[idx] : 0.0;
  }
  for (int r = tid; r < nv; r += nt) {
    wts[W_QV * nv + r] = P.Qv[r * nv + r];
    wts[W_R * nv + r] = P.R[r * nv + r];
    wts[W_QFV * nv + r] = P.Qfv[r * nv + r];
    taus[r] = slab[(size_t)(i - 1) * slab_stride + 3 * bsz + r];
    taus[nv + r] = (i < N) ? slab[(size_t)i * slab_stride + 3 * bsz + r] : 0.0;
    taus[2 * nv + r] = (i < N - 1) ? slab[(size_t)(i + 1) * slab_stride + 3 * bsz + r] : 0.0;
  }
  for (int c = tid; c < nq; c += nt) {
    wq[c] = P.Qq[c * nq + c];
    wfq[c] = P.Qfq[c * nq + c];
    qm1[c] = q[(i - 1) * nq + c];
    q0[c] = q[i * nq + c];
    q1[c] = (i < N) ? q[(i + 1) * nq + c] : 0.0;
  }
  // N+(q_i), N+(q_{i+1}), v_i, v_{i+1}: the finite-difference kernel left them in HBM when it ran
  // over the whole horizon for this q (v_res != nullptr); a rank that only holds a k-range shard
  // of that kernel's outputs recomputes them (same expressions, same bits)
  if (v_res) {
    for (int idx = tid; idx < bsz; idx += nt) {
      N0[idx] = nplus_res[(size_t)i * bsz + idx];
      if (i < N) N1[idx] = nplus_res[(size_t)(i + 1) * bsz + idx];
    }
    for (int r = tid; r < nv; r += nt) {
      v0[r] = v_res[i * nv + r];
      if (i < N) v1[r] = v_res[(i + 1) * nv + r];
    }
  }
  __syncthreads();
  if (stop_after == 1) return;  // (profiling aid: phase timing by truncation)
  if (!v_res) {
    nplus_block(M, q0, N0, tid, nt);
    velocity_block(M, N0, q0, qm1, dt, v0, tid, nt);
    if (i < N) {
      nplus_block(M, q1, N1, tid, nt);
      velocity_block(M, N1, q1, q0, dt, v1, tid, nt);
    }
    __syncthreads();
  }
  if (stop_after == 2) return;
  const double idt = 1 / dt, midt = -1 / dt;
  // weights of the V- and W1-terms depend on the row (TO.cc:1127-1161): Qv, or Qf_v at the end
  const double* wV = wts + ((i < N) ? W_QV : W_QFV) * nv;
  const double* wW1 = wts + ((i < N - 1) ? W_QV : W_QFV) * nv;
  const double* wR = wts + W_R * nv;
  for (int idx = tid; idx < bsz; idx += nt) {
    const int c = idx / nv, l = idx - c * nv, o = c * nvp + l;
    const double vv = idt * N0[idx], w1 = (i < N) ? midt * N1[idx] : 0.0;
    ops[S_V * psz + o] = vv;
    ops[S_W * psz + o] = midt * N0[idx];
    ops[S_W1 * psz + o] = w1;
    ops[X_V * psz + o] = vv * wV[l];
    ops[X_W1 * psz + o] = w1 * wW1[l];
    ops[X_P * psz + o] = ops[S_PM1 * psz + o] * wR[l];
    ops[X_T * psz + o] = ops[S_T0 * psz + o] * wR[l];
    ops[X_M * psz + o] = ops[S_MP1 * psz + o] * wR[l];
  }
  for (int r = tid; r < nv; r += nt) {
    ve[r] = v0[r] - P.v_nom[i * nv + r];
    vep[r] = (i < N) ? v1[r] - P.v_nom[(i + 1) * nv + r] : 0.0;
  }
  for (int c = tid; c < nq; c += nt) qe[c] = q0[c] - P.q_nom[i * nq + c];
  __syncthreads();
  if (stop_after == 3) return;

  // sum_l X(l, r) * B(l, c), l ascending, products and sums rounded separately (no FMA)
  auto term = [&](int xa, int sb, int r, int c) {
    const double* A = ops + xa * psz + r * nvp;
    const double* B = ops + sb * psz + c * nvp;
    const double2* A2 = reinterpret_cast<const double2*>(A);
    const double2* B2 = reinterpret_cast<const double2*>(B);
    double2 a = A2[0], b = B2[0];
    double acc = a.x * b.x;
    if (nv > 1) acc = acc + a.y * b.y;
    const int np = nv >> 1;
#pragma unroll 4
    for (int m = 1; m < np; ++m) {
      a = A2[m]; b = B2[m];
      acc = acc + a.x * b.x;
      acc = acc + a.y * b.y;
    }
    if ((nv & 1) && nv > 1) acc = acc + A[nv - 1] * B[nv - 1];
    return acc;
  };

  if (part == 0) {
    // C_i, lower triangle (TO.cc:1127-1137 / :1157-1161), mirrored (MakeSymmetric)
    const int ntri = nq * (nq + 1) / 2;
    for (int t = tid; t < ntri; t += nt) {
      int c = 0, rem = t;
      while (rem >= nq - c) { rem -= nq - c; ++c; }
      const int r = c + rem;
      double out = (i < N) ? P.Qq[c * nq + r] : P.Qfq[c * nq + r];  // TO.cc:1128 / :1158
      out = out + term(X_V, S_V, r, c);
      out = out + term(X_P, S_PM1, r, c);
      if (i < N) {
        out = out + term(X_T, S_T0, r, c);
        if (i < N - 1) out = out + term(X_M, S_MP1, r, c);
        out = out + term(X_W1, S_W1, r, c);
      }
      Cg[c * nq + r] = out;
      Cg[r * nq + c] = out;
    }
  } else if (part <= 2) {
    // B_i (TO.cc:1140-1147): columns [c_lo, c_hi)
    const int half = (nq + 1) / 2, c_lo = (part == 1) ? 0 : half, c_hi = (part == 1) ? half : nq;
    for (int idx = c_lo * nq + tid; idx < c_hi * nq; idx += nt) {
      const int c = idx / nq, r = idx - c * nq;
      double out = 0.0;
      if (i >= 2) {
        out = term(X_P, S_TM1, r, c);
        if (i < N) out = out + term(X_T, S_M0, r, c);
        out = out + term(X_V, S_W, r, c);
      }
      Bg[idx] = out;
    }
  } else {
    // A_i (TO.cc:1150-1153)
    for (int idx = tid; idx < qq; idx += nt) {
      const int c = idx / nq, r = idx - c * nq;
      Ag[idx] = (i >= 3) ? term(X_P, S_MM1, r, c) : 0.0;
    }
    // gradient block (TO.cc:1046-1080), threads of the last wave
    const int j = tid - (nt - 64);
    if (j >= 0 && j < nq) {
      auto vwm = [&](const double* e, const double* w, int slot) {  // sum_r (e_r w_r) J[r][j]
        const double* J = ops + slot * psz + j * nvp;
        double acc = (e[0] * w[0]) * J[0];
#pragma unroll 6
        for (int r = 1; r < nv; ++r) acc += (e[r] * w[r]) * J[r];
        return acc;
      };
      double gj;
      if (i < N) {
        gj = qe[j] * wq[j];
        gj = gj + vwm(ve, wts + W_QV * nv, S_V);
        gj = gj + vwm(vep, wts + ((i == N - 1) ? W_QFV : W_QV) * nv, S_W1);
        gj = gj + vwm(taus, wts + W_R * nv, S_PM1);
        gj = gj + vwm(taus + nv, wts + W_R * nv, S_T0);
        if (i != N - 1) gj = gj + vwm(taus + 2 * nv, wts + W_R * nv, S_MP1);
      } else {
        gj = vwm(taus, wts + W_R * nv, S_PM1);
        gj = gj + qe[j] * wfq[j];
        gj = gj + vwm(ve, wts + W_QFV * nv, S_V);
      }
      g[(size_t)i * nq + j] = gj;
    }
  }
}

__global__ void __launch_bounds__(256)
assemble_diag_kernel(DevModel M, DevProblem P, const double* __restrict__ q, const double* __restrict__ slab,
                     int slab_stride, double* __restrict__ g, double* __restrict__ HA, double* __restrict__ HB,
                     double* __restrict__ HC, int stop_after, const double* __restrict__ v_res,
                     const double* __restrict__ nplus_res, size_t pstride, const double* __restrict__ gate, AltSel alt) {
  if (gate && *at_problem(gate, (size_t)blockIdx.z * pstride) == 0.0) return;   // (idto_hip_tr_solve: this problem's step was rejected, its g and H stay)
  const size_t o = (size_t)blockIdx.z * pstride;  // problem of the batch
  const size_t w = o + (size_t)alt_offset(alt, o);   // ... and the iterate's set of fd_kernel outputs
  assemble_diag_body(M, at_problem(P, o), at_problem(q, o), at_problem(slab, w), slab_stride, at_problem(g, o),
                     at_problem(HA, o), at_problem(HB, o), at_problem(HC, o), stop_after,
                     v_res ? at_problem(v_res, w) : nullptr, nplus_res ? at_problem(nplus_res, w) : nullptr,
                     (int)blockIdx.x, (int)blockIdx.y);
}

// ---------------------------------------------------------------------------
// assemble_terms_kernel: block row i of g and H from the per-record products the finite-difference
// workgroups left in HBM (asm_terms_stride) plus the velocity terms (N+ / dt, formed here: they
// are sparse and cheap).  Same sums in the same order as assemble_diag_kernel - the entries of CP,
// CT, ... ARE its term() values - so the results are bit-identical; what is gone is the staging of
// six dtau/dq blocks per workgroup, four workgroups per block row (3.1 MB of fetches for 0.7 MB of
// data, and 12 us of mostly operand staging on the critical path of the iteration).
// Grid (N + 1, 4, batch): part 0 C_i and the gradient block, 1 / 2 the column halves of B_i, 3 A_i.
// (WT: results leave with agent-scope write-through stores - the form in which another workgroup of the SAME launch
// may consume them, see penta_pipe.h; bits are the same either way)
template <bool WT>
IDTO_DEV void asm_put(double* p, double v) {
  if (WT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}

// The banded KKT system of the equality-constrained step (kkt.h: blocks of K = nq + nu, M_{t,t-2}, M_{t,t-1}, M_{t,t} and the
// right-hand side [g_t ; h_{t-1}]) written by the assembly as it goes: every entry of g and H it produces goes into
// the KKT bands too, part 3's workgroup adds the rows of J (the records of step t - 1), the dummy multiplier's identity
// and h - kkt_build_kernel's entries, whose launch (5.7 us of hopper's iteration, 11 of allegro's) this saves.  KC == nullptr: off.
struct KktSink {
  double *KA, *KB, *KC, *rhs;
  int K, nu;
  const double* slab; int slab_stride;   // fd_kernel's records of the iterate the assembly runs for
  const int* dofs;
};

// block row i, part `part` (all threads of the workgroup; `lds`: asm_terms_lds bytes)
template <bool WT>
__device__ __forceinline__ void assemble_terms_row(int nq, int nv, const DevProblem& P, const double* __restrict__ q,
                                                   const double* __restrict__ terms, const double* __restrict__ v_res,
                                                   const double* __restrict__ nplus_res, double* __restrict__ g,
                                                   double* __restrict__ HA, double* __restrict__ HB, double* __restrict__ HC,
                                                   int i, int part, double* lds, const KktSink S = KktSink{}) {
  const int tid = threadIdx.x, nt = blockDim.x;
  const int N = P.N;
  const int bsz = nv * nq, qq = nq * nq;
  const int nvp = (nv + 1) & ~1, psz = nvp * nq;
  const double dt = P.dt;
  double* Cg = HC + (size_t)i * qq;
  double* Bg = HB + (size_t)i * qq;
  double* Ag = HA + (size_t)i * qq;
  const int KS = S.K, kks = KS * KS;
  double* Ck = S.KC ? S.KC + (size_t)i * kks : nullptr;
  double* Bk = S.KC ? S.KB + (size_t)i * kks : nullptr;
  double* Ak = S.KC ? S.KA + (size_t)i * kks : nullptr;
  if (S.KC && part == 3) {
    // the entries of the KKT block row that are not H's (kkt_build_kernel's rules: the rows of J against block column
    // sc = i - 2 + band come from record i - 1's block `band`, q_0 is no variable; the transposed entries are the diagonal
    // block's only; mu_0 is a dummy with an identity block), and h = tau_{i-1}[dof]
    const double* rec = (i >= 1) ? S.slab + (size_t)(i - 1) * S.slab_stride : S.slab;
    for (int e = tid; e < 3 * kks; e += nt) {
      const int band = e / kks, ee = e - band * kks, c = ee / KS, r = ee - c * KS, sc = i - 2 + band;
      if (r < nq && c < nq) continue;   // (H's entries: the parts that form them write them)
      double v = 0.0;
      if (sc >= 0) {
        if (r >= nq && c < nq) {
          const bool zero = i < 1 || (band == 0 && i - 1 < 2) || (band == 1 && i - 1 < 1);
          if (!zero) v = rec[(size_t)band * nv * nq + c * nv + S.dofs[r - nq]];
        } else if (r < nq) {
          if (band == 2 && i >= 1) v = rec[(size_t)2 * nv * nq + r * nv + S.dofs[c - nq]];
        } else if (i == 0 && band == 2 && r == c) {
          v = 1.0;
        }
      }
      (band == 0 ? Ak : band == 1 ? Bk : Ck)[ee] = v;
    }
    for (int r = nq + tid; r < KS; r += nt)
      S.rhs[(size_t)i * KS + r] = (i >= 1) ? S.slab[(size_t)(i - 1) * S.slab_stride + 3 * nv * nq + S.dofs[r - nq]] : 0.0;
  }
  if (i == 0) {
    if (part == 0) {
      for (int idx = tid; idx < qq; idx += nt) {
        asm_put<WT>(Cg + idx, (idx / nq == idx % nq) ? 1.0 : 0.0);
        asm_put<WT>(Bg + idx, 0.0);
        asm_put<WT>(Ag + idx, 0.0);
        if (Ck) {   // (kkt_build_kernel: block columns before the first do not exist - zeros)
          const int c = idx / nq, r = idx - c * nq;
          Ck[c * KS + r] = (c == r) ? 1.0 : 0.0; Bk[c * KS + r] = 0.0; Ak[c * KS + r] = 0.0;
        }
      }
      for (int j = tid; j < nq; j += nt) { asm_put<WT>(g + j, 0.0); if (Ck) S.rhs[j] = 0.0; }
    }
    return;
  }
  const int ts = asm_terms_stride(nq);
  const double* Tm1 = terms + (size_t)(i - 1) * ts;            // record i-1
  const double* T0 = terms + (size_t)(i < N ? i : 0) * ts;     // record i   (i < N)
  const double* Tp1 = terms + (size_t)(i < N - 1 ? i + 1 : 0) * ts;
  if (part == 3) {   // A_i (TO.cc:1150-1153): P_{i-1}^T R' M_{i-1}, nothing to add
    for (int e = tid; e < qq; e += nt) {
      const double out = (i >= 3) ? Tm1[5 * qq + e] : 0.0;
      asm_put<WT>(Ag + e, out);
      if (Ak) { const int c = e / nq; Ak[c * KS + e - c * nq] = out; }
    }
    return;
  }
  // The product terms of the thread's FIRST item (its only one when the workgroup has a thread per item) are requested
  // here, before the staging below waits for ITS loads: two memory latencies in a row become one.  (Values and the
  // order of the sums are what they were.)
  const int ntri = nq * (nq + 1) / 2;
  double pre0 = 0.0, pre1 = 0.0, pre2 = 0.0, pre3 = 0.0, pre4 = 0.0;
  if (part == 0) {
    if (tid < ntri) {
      int c = 0, rem = tid;
      while (rem >= nq - c) { rem -= nq - c; ++c; }
      const int e = c * nq + c + rem;
      pre0 = Tm1[e];
      pre1 = (i < N) ? T0[qq + e] : 0.0;
      pre2 = (i < N - 1) ? Tp1[2 * qq + e] : 0.0;
      pre3 = (i < N) ? P.Qq[e] : P.Qfq[e];
    } else if (tid < ntri + nq) {
      const int j = tid - ntri;
      pre0 = Tm1[6 * qq + j];
      pre1 = (i < N) ? T0[6 * qq + nq + j] : 0.0;
      pre2 = (i < N - 1) ? Tp1[6 * qq + 2 * nq + j] : 0.0;
      pre3 = q[i * nq + j];
      pre4 = P.q_nom[i * nq + j];
    }
  } else {
    const int half0 = (nq + 1) / 2, e0 = ((part == 1) ? 0 : half0) * nq + tid;
    if (i >= 2 && e0 < ((part == 1) ? half0 : nq) * nq) {
      pre0 = Tm1[3 * qq + e0];
      pre1 = (i < N) ? T0[4 * qq + e0] : 0.0;
    }
  }
  // one round of loads: everything this part needs comes straight from HBM / L2 into its LDS form
  enum { S_V = 0, S_W, S_W1, X_V, X_W1, S_COUNT };
  double* ops = lds;                 // S_COUNT * psz
  double* ve = ops + S_COUNT * psz;  // (v_i - vnom_i) * weight        (gradient)
  double* vep = ve + nv;             // (v_{i+1} - vnom_{i+1}) * weight
  const double idt = 1 / dt, midt = -1 / dt;
  // weights of the V- and W1-terms depend on the row (TO.cc:1127-1161): Qv, or Qf_v at the end
  const double* wV = (i < N) ? P.Qv : P.Qfv;
  const double* wW1 = (i < N - 1) ? P.Qv : P.Qfv;
  for (int idx = tid; idx < bsz; idx += nt) {
    const int c = idx / nv, l = idx - c * nv, o = c * nvp + l;
    const double n0 = nplus_res[(size_t)i * bsz + idx];
    const double vv = idt * n0;
    ops[S_V * psz + o] = vv;
    if (part == 0) {
      const double w1 = (i < N) ? midt * nplus_res[(size_t)(i + 1) * bsz + idx] : 0.0;
      ops[S_W1 * psz + o] = w1;
      ops[X_W1 * psz + o] = w1 * wW1[l * nv + l];
      ops[X_V * psz + o] = vv * wV[l * nv + l];
    } else {
      ops[S_W * psz + o] = midt * n0;
      ops[X_V * psz + o] = vv * wV[l * nv + l];
    }
  }
  double* wdv = vep + nv;            // diagonal of the V-term's weights as the gradient uses them ...
  double* wdw = wdv + nv;            // ... and of the W1-term's  (staged: the gradient's loop over r fetched them
                                     // one global load per step, 18 round trips in a row on the slowest part)
  if (part == 0) {
    const double* WVd = (i < N) ? P.Qv : P.Qfv;
    const double* WWd = (i == N - 1) ? P.Qfv : P.Qv;
    for (int r = tid; r < nv; r += nt) {
      ve[r] = v_res[i * nv + r] - P.v_nom[i * nv + r];
      vep[r] = (i < N) ? v_res[(i + 1) * nv + r] - P.v_nom[(i + 1) * nv + r] : 0.0;
      wdv[r] = WVd[r * nv + r];
      wdw[r] = WWd[r * nv + r];
    }
  }
  __syncthreads();
  auto term = [&](int xa, int sb, int r, int c) { return asm_dot(ops + xa * psz + r * nvp, ops + sb * psz + c * nvp, nv); };
  if (part == 0) {
    // C_i, lower triangle (TO.cc:1127-1137 / :1157-1161), mirrored (MakeSymmetric); then the gradient
    for (int item = tid; item < ntri + nq; item += nt) {
      const bool first = item == tid;
      if (item < ntri) {
        int c = 0, rem = item;
        while (rem >= nq - c) { rem -= nq - c; ++c; }
        const int r = c + rem, e = c * nq + r;
        const double w0 = first ? pre3 : ((i < N) ? P.Qq[e] : P.Qfq[e]);
        const double tP = first ? pre0 : Tm1[e];                                   // P_{i-1}^T R' P_{i-1}
        const double tT = first ? pre1 : ((i < N) ? T0[qq + e] : 0.0);             // T_i^T R' T_i
        const double tM = first ? pre2 : ((i < N - 1) ? Tp1[2 * qq + e] : 0.0);    // M_{i+1}^T R' M_{i+1}
        const double* const A2[2] = {ops + X_V * psz + r * nvp, ops + X_W1 * psz + r * nvp};
        const double* const B2[2] = {ops + S_V * psz + c * nvp, ops + S_W1 * psz + c * nvp};
        double tv[2];   // the V- and the W1-term, two chains in lockstep (W1 operands are zeros at i == N)
        asm_dot_n<2>(A2, B2, nv, tv);
        double out = w0;
        out = out + tv[0];
        out = out + tP;
        if (i < N) {
          out = out + tT;
          if (i < N - 1) out = out + tM;
          out = out + tv[1];
        }
        asm_put<WT>(Cg + e, out);
        asm_put<WT>(Cg + r * nq + c, out);
        if (Ck) { Ck[c * KS + r] = out; Ck[r * KS + c] = out; }
      } else {   // gradient block (TO.cc:1046-1080)
        const int j = item - ntri;
        const double gP = first ? pre0 : Tm1[6 * qq + j];                              // P_{i-1}^T R' tau_{i-1}
        const double gT = first ? pre1 : ((i < N) ? T0[6 * qq + nq + j] : 0.0);        // T_i^T R' tau_i
        const double gM = first ? pre2 : ((i < N - 1) ? Tp1[6 * qq + 2 * nq + j] : 0.0);   // M_{i+1}^T R' tau_{i+1}
        const double qej = first ? pre3 - pre4 : q[i * nq + j] - P.q_nom[i * nq + j];
        const double wq = (i < N) ? P.Qq[j * nq + j] : P.Qfq[j * nq + j];
        // sum_r (e_r w_r) J[r][j] for the V- and the W1-term, two chains in lockstep
        const double* JV = ops + S_V * psz + j * nvp;
        const double* JW = ops + S_W1 * psz + j * nvp;
        double gv = (ve[0] * wdv[0]) * JV[0], gw = (vep[0] * wdw[0]) * JW[0];
        for (int r = 1; r < nv; ++r) {
          gv += (ve[r] * wdv[r]) * JV[r];
          gw += (vep[r] * wdw[r]) * JW[r];
        }
        double gj;
        if (i < N) {
          gj = qej * wq;
          gj = gj + gv;
          gj = gj + gw;
          gj = gj + gP;
          gj = gj + gT;
          if (i != N - 1) gj = gj + gM;
        } else {
          gj = gP;
          gj = gj + qej * wq;
          gj = gj + gv;
        }
        asm_put<WT>(g + (size_t)i * nq + j, gj);
        if (Ck) S.rhs[(size_t)i * KS + j] = gj;
      }
    }
  } else {
    // B_i (TO.cc:1140-1147): columns [c_lo, c_hi)
    const int half = (nq + 1) / 2, c_lo = (part == 1) ? 0 : half, c_hi = (part == 1) ? half : nq;
    for (int e = c_lo * nq + tid; e < c_hi * nq; e += nt) {
      const int c = e / nq, r = e - c * nq;
      double out = 0.0;
      if (i >= 2) {
        const bool first = e == c_lo * nq + tid;
        const double tPT = first ? pre0 : Tm1[3 * qq + e];                        // P_{i-1}^T R' T_{i-1}
        const double tTM = first ? pre1 : ((i < N) ? T0[4 * qq + e] : 0.0);       // T_i^T R' M_i
        out = tPT;
        if (i < N) out = out + tTM;
        out = out + term(X_V, S_W, r, c);
      }
      asm_put<WT>(Bg + e, out);
      if (Bk) Bk[c * KS + r] = out;
    }
  }
}

__global__ void __launch_bounds__(512)
assemble_terms_kernel(DevModel M, DevProblem P, const double* __restrict__ q, const double* __restrict__ terms,
                      const double* __restrict__ v_res, const double* __restrict__ nplus_res, double* __restrict__ g,
                      double* __restrict__ HA, double* __restrict__ HB, double* __restrict__ HC, size_t pstride,
                      const double* __restrict__ gate, AltSel alt, KktSink S, size_t kstride) {
  if (gate && *at_problem(gate, (size_t)blockIdx.z * pstride) == 0.0) return;   // (idto_hip_tr_solve: this problem's step was rejected, its g and H stay - and so does its KKT system)
  {
    const size_t o = (size_t)blockIdx.z * pstride, w = o + (size_t)alt_offset(alt, o);
    P = at_problem(P, o); q = at_problem(q, o); terms = at_problem(terms, w); v_res = at_problem(v_res, w);
    nplus_res = at_problem(nplus_res, w); g = at_problem(g, o);
    HA = at_problem(HA, o); HB = at_problem(HB, o); HC = at_problem(HC, o);
    if (S.KC) {
      const size_t ok = (size_t)blockIdx.z * kstride;
      S.KA = at_problem(S.KA, ok); S.KB = at_problem(S.KB, ok); S.KC = at_problem(S.KC, ok); S.rhs = at_problem(S.rhs, ok);
      S.slab = at_problem(S.slab, w);
    }
  }
  extern __shared__ double lds[];
  assemble_terms_row<false>(M.nq, M.nv, P, q, terms, v_res, nplus_res, g, HA, HB, HC, (int)blockIdx.x, (int)blockIdx.y, lds, S);
}

// ---------------------------------------------------------------------------
// penta_kernel (v1): block-Thomas factorisation of the symmetric block
// penta-diagonal H (lower bands A, B, C in HBM) fused with the solve of one
// right-hand side.  Restates penta_diagonal_solver.h:124-248 with the per-block
// LU (partial pivoting, first maximum) of oracle/penta.h.  One workgroup; the
// factors (K, LU(G), pivots, Y, Z) are also stored for later multi-RHS solves.
//   rhs_sign: +1 rhs = b ; -1 rhs = -b (the Gauss-Newton step solves H p = -g).
IDTO_DEV void blk_sub_mul(const double* X, const double* L, const double* R, double* out, int k, int tid, int nt) {
  for (int idx = tid; idx < k * k; idx += nt) {
    const int c = idx / k, r = idx - c * k;
    double acc = L[r] * R[c * k];
    for (int j = 1; j < k; ++j) acc += L[j * k + r] * R[c * k + j];
    out[idx] = X[idx] - acc;
  }
}

__global__ void penta_kernel(int n, int k, const double* __restrict__ HA, const double* __restrict__ HB,
                             const double* __restrict__ HC, const double* __restrict__ b, double rhs_sign,
                             double* __restrict__ x, double* __restrict__ Kst, double* __restrict__ LUst,
                             int* __restrict__ pivst, double* __restrict__ Yst, double* __restrict__ Zst,
                             unsigned* __restrict__ status, unsigned fact_id, size_t pstride) {
  {
    const size_t o = (size_t)blockIdx.y * pstride;
    HA = at_problem(HA, o); HB = at_problem(HB, o); HC = at_problem(HC, o); b = at_problem(b, o); x = at_problem(x, o);
    Kst = at_problem(Kst, o); LUst = at_problem(LUst, o); pivst = at_problem(pivst, o); Yst = at_problem(Yst, o);
    Zst = at_problem(Zst, o);
    status += 2 * blockIdx.y;
  }
  extern __shared__ double lds[];
  const int tid = threadIdx.x, nt = blockDim.x;
  const int kk = k * k;
  const int ncols = 3 * k + 1;
  double* Ai = lds;
  double* Bi = Ai + kk;
  double* Ci = Bi + kk;
  double* Di = Ci + kk;
  double* Ki = Di + kk;
  double* Tm = Ki + kk;
  double* Ym2 = Tm + kk;
  double* Ym1 = Ym2 + kk;
  double* Zm2 = Ym1 + kk;
  double* Zm1 = Zm2 + kk;
  double* W = Zm1 + kk;              // k x ncols, column-major: [G | Y | Z | r]
  double* rv = W + k * ncols;        // (n+2) k : r_{i-2} storage with the +2 offset of the reference
  double* tvec = rv + (n + 2) * k;   // k
  int* ipiv = (int*)(tvec + k);      // k ints

  for (int idx = tid; idx < kk; idx += nt) { Ym2[idx] = 0; Ym1[idx] = 0; Zm2[idx] = 0; Zm1[idx] = 0; }
  for (int idx = tid; idx < (n + 2) * k; idx += nt)
    rv[idx] = (idx < 2 * k) ? 0.0 : rhs_sign * b[idx - 2 * k];
  __syncthreads();

  for (int i = 0; i < n; ++i) {
    double* G = W;
    double* Yw = W + kk;
    double* Zw = W + 2 * kk;
    double* rw = W + 3 * kk;
    for (int idx = tid; idx < kk; idx += nt) {
      const int c = idx / k, r = idx - c * k;
      Ai[idx] = HA[(size_t)i * kk + idx];
      Bi[idx] = HB[(size_t)i * kk + idx];
      Ci[idx] = HC[(size_t)i * kk + idx];
      Di[idx] = (i < n - 1) ? HB[(size_t)(i + 1) * kk + r * k + c] : 0.0;  // D_i = B_{i+1}^T
      Zw[idx] = (i < n - 2) ? HA[(size_t)(i + 2) * kk + r * k + c] : 0.0;  // E_i = A_{i+2}^T
    }
    __syncthreads();
    blk_sub_mul(Bi, Ai, Ym2, Ki, k, tid, nt);   // K = B - A Y_{i-2}
    blk_sub_mul(Ci, Ai, Zm2, Tm, k, tid, nt);   // G' = C - A Z_{i-2}
    for (int r = tid; r < k; r += nt) {         // r_i -= A r_{i-2}
      const double* rim2 = rv + i * k;
      double acc = Ai[r] * rim2[0];
      for (int c = 1; c < k; ++c) acc += Ai[c * k + r] * rim2[c];
      tvec[r] = rv[(i + 2) * k + r] - acc;
    }
    __syncthreads();
    blk_sub_mul(Tm, Ki, Ym1, G, k, tid, nt);    // G = G' - K Y_{i-1}
    blk_sub_mul(Di, Ki, Zm1, Yw, k, tid, nt);   // Y' = D - K Z_{i-1}
    for (int r = tid; r < k; r += nt) {         // r_i -= K r_{i-1}
      const double* rim1 = rv + (i + 1) * k;
      double acc = Ki[r] * rim1[0];
      for (int c = 1; c < k; ++c) acc += Ki[c * k + r] * rim1[c];
      rw[r] = tvec[r] - acc;
    }
    __syncthreads();
    // LU with partial pivoting of G, elimination applied to [Y' | E | r] as well
    for (int j = 0; j < k; ++j) {
      int p = j;
      double best = __builtin_fabs(G[j * k + j]);
      for (int r = j + 1; r < k; ++r) {
        const double a = __builtin_fabs(G[j * k + r]);
        if (a > best) { best = a; p = r; }
      }
      if (tid == 0) ipiv[j] = p;
      __syncthreads();
      if (p != j)
        for (int c = tid; c < ncols; c += nt) {
          const double t = W[c * k + j];
          W[c * k + j] = W[c * k + p];
          W[c * k + p] = t;
        }
      __syncthreads();
      const double d = G[j * k + j];
      // Eigen's PartialPivLU "always succeeds" (penta_diagonal_solver.h:108-110); a pivot that is
      // exactly zero or not finite after pivoting (singular H) is still reported to the host
      if (tid == 0 && !(__builtin_fabs(d) > 0.0 && __builtin_fabs(d) < __builtin_inf())) {
        __hip_atomic_store(status, fact_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_fetch_add(status + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      for (int r = j + 1 + tid; r < k; r += nt) G[j * k + r] = G[j * k + r] / d;
      __syncthreads();
      const int nr = k - j - 1, nc = ncols - j - 1;
      for (int idx = tid; idx < nr * nc; idx += nt) {
        const int cc = idx / nr, rr = idx - cc * nr;
        const int c = j + 1 + cc, r = j + 1 + rr;
        W[c * k + r] = W[c * k + r] - G[j * k + r] * W[c * k + j];
      }
      __syncthreads();
    }
    // back substitution, one thread per right-hand-side column
    for (int c = k + tid; c < ncols; c += nt) {
      double* xc = W + c * k;
      for (int j = k - 1; j >= 0; --j) {
        xc[j] = xc[j] / G[j * k + j];
        const double xj = xc[j];
        for (int r = 0; r < j; ++r) xc[r] = xc[r] - G[j * k + r] * xj;
      }
    }
    __syncthreads();
    // store the factors, rotate Y/Z
    for (int idx = tid; idx < kk; idx += nt) {
      Kst[(size_t)i * kk + idx] = Ki[idx];
      LUst[(size_t)i * kk + idx] = G[idx];
      Yst[(size_t)i * kk + idx] = Yw[idx];
      Zst[(size_t)i * kk + idx] = Zw[idx];
      Ym2[idx] = Ym1[idx];
      Zm2[idx] = Zm1[idx];
    }
    for (int r = tid; r < k; r += nt) {
      rv[(i + 2) * k + r] = rw[r];
      pivst[i * k + r] = ipiv[r];
    }
    __syncthreads();
    for (int idx = tid; idx < kk; idx += nt) { Ym1[idx] = Yw[idx]; Zm1[idx] = Zw[idx]; }
    __syncthreads();
  }

  // backward substitution (penta_diagonal_solver.h:228-247): x lives in rv (+2 offset)
  for (int i = n - 2; i >= 0; --i) {
    const double* Yi = Yst + (size_t)i * kk;
    const double* Zi = Zst + (size_t)i * kk;
    const double* xp1 = rv + (i + 3) * k;
    const double* xp2 = rv + (i + 4) * k;
    for (int r = tid; r < k; r += nt) {
      double acc = Yi[r] * xp1[0];
      for (int c = 1; c < k; ++c) acc += Yi[c * k + r] * xp1[c];
      double val = rv[(i + 2) * k + r] - acc;
      if (i <= n - 3) {
        double acc2 = Zi[r] * xp2[0];
        for (int c = 1; c < k; ++c) acc2 += Zi[c * k + r] * xp2[c];
        val = val - acc2;
      }
      tvec[r] = val;
    }
    __syncthreads();
    for (int r = tid; r < k; r += nt) rv[(i + 2) * k + r] = tvec[r];
    __syncthreads();
  }
  for (int idx = tid; idx < n * k; idx += nt) x[idx] = rv[2 * k + idx];
}

// Multi-RHS solve with the stored factors: block <-> right-hand side.
__global__ void penta_solve_kernel(int n, int k, const double* __restrict__ HA, const double* __restrict__ Kst,
                                   const double* __restrict__ LUst, const int* __restrict__ pivst,
                                   const double* __restrict__ Yst, const double* __restrict__ Zst,
                                   const double* __restrict__ b, double* __restrict__ x) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x, nt = blockDim.x;
  const int kk = k * k;
  double* rv = lds;                 // (n+2) k
  double* tvec = rv + (n + 2) * k;  // k
  const double* bb = b + (size_t)blockIdx.x * n * k;
  double* xx = x + (size_t)blockIdx.x * n * k;
  for (int idx = tid; idx < (n + 2) * k; idx += nt) rv[idx] = (idx < 2 * k) ? 0.0 : bb[idx - 2 * k];
  __syncthreads();
  for (int i = 0; i < n; ++i) {
    const double* Ai = HA + (size_t)i * kk;
    const double* Ki = Kst + (size_t)i * kk;
    const double* LU = LUst + (size_t)i * kk;
    const int* piv = pivst + i * k;
    for (int r = tid; r < k; r += nt) {
      const double* rim2 = rv + i * k;
      const double* rim1 = rv + (i + 1) * k;
      double acc = Ai[r] * rim2[0];
      for (int c = 1; c < k; ++c) acc += Ai[c * k + r] * rim2[c];
      double val = rv[(i + 2) * k + r] - acc;
      double acc2 = Ki[r] * rim1[0];
      for (int c = 1; c < k; ++c) acc2 += Ki[c * k + r] * rim1[c];
      tvec[r] = val - acc2;
    }
    __syncthreads();
    if (tid == 0) {  // G^{-1} via the stored LU: swaps, unit-lower solve, upper solve
      for (int j = 0; j < k; ++j)
        if (piv[j] != j) { const double t = tvec[j]; tvec[j] = tvec[piv[j]]; tvec[piv[j]] = t; }
      for (int j = 0; j < k; ++j) {
        const double xj = tvec[j];
        for (int r = j + 1; r < k; ++r) tvec[r] = tvec[r] - LU[j * k + r] * xj;
      }
      for (int j = k - 1; j >= 0; --j) {
        tvec[j] = tvec[j] / LU[j * k + j];
        const double xj = tvec[j];
        for (int r = 0; r < j; ++r) tvec[r] = tvec[r] - LU[j * k + r] * xj;
      }
    }
    __syncthreads();
    for (int r = tid; r < k; r += nt) rv[(i + 2) * k + r] = tvec[r];
    __syncthreads();
  }
  for (int i = n - 2; i >= 0; --i) {
    const double* Yi = Yst + (size_t)i * kk;
    const double* Zi = Zst + (size_t)i * kk;
    const double* xp1 = rv + (i + 3) * k;
    const double* xp2 = rv + (i + 4) * k;
    for (int r = tid; r < k; r += nt) {
      double acc = Yi[r] * xp1[0];
      for (int c = 1; c < k; ++c) acc += Yi[c * k + r] * xp1[c];
      double val = rv[(i + 2) * k + r] - acc;
      if (i <= n - 3) {
        double acc2 = Zi[r] * xp2[0];
        for (int c = 1; c < k; ++c) acc2 += Zi[c * k + r] * xp2[c];
        val = val - acc2;
      }
      tvec[r] = val;
    }
    __syncthreads();
    for (int r = tid; r < k; r += nt) rv[(i + 2) * k + r] = tvec[r];
    __syncthreads();
  }
  for (int idx = tid; idx < n * k; idx += nt) xx[idx] = rv[2 * k + idx];
}

// probe of device arithmetic (tests/test_gpu_math.py)
__global__ void math_probe_kernel(const double* x, int n, double* sq, double* rc, double* sn, double* cs, double* ex,
                                  double* lg) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i];
  sq[i] = __builtin_sqrt(__builtin_fabs(v));
  rc[i] = 1.0 / v;
  double s, c;
  idto::detmath::sincos(v, &s, &c);
  sn[i] = s;
  cs[i] = c;
  ex[i] = idto::detmath::exp(v);
  lg[i] = idto::detmath::log(__builtin_fabs(v));
}

}  // namespace idto_dev
