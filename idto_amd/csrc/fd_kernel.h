// fd_kernel.h — the finite-difference kernel of one Gauss-Newton iteration on gfx950 (a1-a9 of SURVEY.md §8a):
// N+ / v / a / tau, the partials of the inverse dynamics by forward or central differences (reference
// optimizer/trajectory_optimizer.cc:426-885) and the single-record products of the assembly (:1021-1165).
// A header of its own so that csrc/fd_launch.hip compiles the kernel's instantiations as a translation unit
// of their own (in parallel with the solvers, and with its own scheduling flags); kernels.h includes it for
// the fused launch (fused.h), which embeds fd_body.
#pragma once

#include "batch.h"
#include "id_eval.h"
#include "id_fast.h"

namespace idto_dev {

struct DevProblem {
  int N;
  double dt;
  const double* v_init;  // nv
  const double* q_nom;   // (N+1) nq
  const double* v_nom;   // (N+1) nv
  // weights pre-scaled on the host exactly as reference TO.cc:1103-1107:
  const double* Qq;   // (2 Qq) dt     nq x nq column-major
  const double* Qv;   // (2 Qv) dt     nv x nv
  const double* R;    // (2 R) dt
  const double* Qfq;  // 2 Qf_q
  const double* Qfv;  // 2 Qf_v
  // unscaled weights for the cost (TO.cc:147-176)
  const double* Qq0; const double* Qv0; const double* R0; const double* Qfq0; const double* Qfv0;
};

__host__ __device__ __forceinline__ DevProblem at_problem(DevProblem P, size_t o) {
  P.v_init = at_problem(P.v_init, o); P.q_nom = at_problem(P.q_nom, o); P.v_nom = at_problem(P.v_nom, o);
  P.Qq = at_problem(P.Qq, o); P.Qv = at_problem(P.Qv, o); P.R = at_problem(P.R, o); P.Qfq = at_problem(P.Qfq, o);
  P.Qfv = at_problem(P.Qfv, o);
  P.Qq0 = at_problem(P.Qq0, o); P.Qv0 = at_problem(P.Qv0, o); P.R0 = at_problem(P.R0, o); P.Qfq0 = at_problem(P.Qfq0, o);
  P.Qfv0 = at_problem(P.Qfv0, o);
  return P;
}

// ---------------------------------------------------------------------------
// N+(q) for one configuration into LDS (nv x nq column-major); restates
// oracle Dynamics::Nplus / reference TO.cc:1633-1647.  Called by all threads.
IDTO_DEV void nplus_block(const DevModel& M, const double* q, double* Nout, int tid, int nthreads) {
  const int sz = M.nv * M.nq;
  for (int i = tid; i < sz; i += nthreads) Nout[i] = 0.0;
  __syncthreads();
  for (int b = tid; b < M.nb; b += nthreads) {
    const int qs = M.qstart[b], vs = M.vstart[b], jt = M.jtype[b], nv = M.nv;
    if (jt == IDTO_JOINT_REVOLUTE || jt == IDTO_JOINT_PRISMATIC) {
      Nout[qs * nv + vs] = 1.0;
    } else if (jt == IDTO_JOINT_PLANAR) {
      for (int k = 0; k < 3; ++k) Nout[(qs + k) * nv + vs + k] = 1.0;
    } else {
      const double* qq = q + qs;
      const double nrm = __builtin_sqrt(((qq[0] * qq[0] + qq[1] * qq[1]) + qq[2] * qq[2]) + qq[3] * qq[3]);
      const double t0 = qq[0] / nrm, t1 = qq[1] / nrm, t2 = qq[2] / nrm, t3 = qq[3] / nrm;
      const double t[4] = {t0, t1, t2, t3};
      const double w2 = 2.0 * t0, x2 = 2.0 * t1, y2 = 2.0 * t2, z2 = 2.0 * t3;
      const double LT[3][4] = {{-x2, w2, -z2, y2}, {-y2, z2, w2, -x2}, {-z2, -y2, x2, w2}};
      double D[4][4];
      for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) D[r][c] = ((r == c ? 1.0 : 0.0) - t[r] * t[c]) / nrm;
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) {
          double acc = LT[r][0] * D[0][c];
          for (int k = 1; k < 4; ++k) acc += LT[r][k] * D[k][c];
          Nout[(qs + c) * nv + vs + r] = acc;
        }
      for (int k = 0; k < 3; ++k) Nout[(qs + 4 + k) * nv + vs + 3 + k] = 1.0;
    }
  }
  __syncthreads();
}

// N+ of TWO configurations at once (fd_kernel needs q_k and q_{k+1}): wavefront 0 works on the
// first, wavefront 1 on the second (a single wavefront does them one after the other); within a
// wavefront lanes 0..47 take the bodies, lanes 48..59 the 3x4 quaternion block entry by entry
// (each entry is 9 IEEE divisions deep instead of 21 in sequence).  Same expressions as
// nplus_block, hence the same bits.
// `zeroed`: the caller cleared both blocks behind a barrier of its own (fd_body: together with its first loads)
IDTO_DEV void nplus_pair(const DevModel& M, const double* qa, double* Na, const double* qb, double* Nb, int tid,
                         int nthreads, bool zeroed = false) {
  const int sz = M.nv * M.nq, nv = M.nv;
  if (!zeroed) {
    for (int i = tid; i < sz; i += nthreads) { Na[i] = 0.0; Nb[i] = 0.0; }
    __syncthreads();
  }
  const int l = tid & 63, grp = tid >> 6, ngrp = (nthreads >= 128) ? 2 : 1;
  for (int cfg = grp; cfg < 2; cfg += ngrp) {
    if (grp >= 2) break;
    const double* q = cfg ? qb : qa;
    double* Nout = cfg ? Nb : Na;
    if (l < 48) {
      for (int b = l; b < M.nb; b += 48) {
        const int qs = M.qstart[b], vs = M.vstart[b], jt = M.jtype[b];
        if (jt == IDTO_JOINT_REVOLUTE || jt == IDTO_JOINT_PRISMATIC) {
          Nout[qs * nv + vs] = 1.0;
        } else if (jt == IDTO_JOINT_PLANAR) {
          for (int k = 0; k < 3; ++k) Nout[(qs + k) * nv + vs + k] = 1.0;
        } else {
          for (int k = 0; k < 3; ++k) Nout[(qs + 4 + k) * nv + vs + 3 + k] = 1.0;
        }
      }
    } else if (l < 60) {
      const int e = l - 48, r = e >> 2, c = e & 3;
      for (int b = 0; b < M.nb; ++b) {
        if (M.jtype[b] != IDTO_JOINT_FLOATING) continue;
        const int qs = M.qstart[b], vs = M.vstart[b];
        const double* qq = q + qs;
        const double nrm = __builtin_sqrt(((qq[0] * qq[0] + qq[1] * qq[1]) + qq[2] * qq[2]) + qq[3] * qq[3]);
        const double t0 = qq[0] / nrm, t1 = qq[1] / nrm, t2 = qq[2] / nrm, t3 = qq[3] / nrm;
        const double w2 = 2.0 * t0, x2 = 2.0 * t1, y2 = 2.0 * t2, z2 = 2.0 * t3;
        // row r of LT = L(2 q~)^T
        const double l0 = (r == 0) ? -x2 : ((r == 1) ? -y2 : -z2);
        const double l1 = (r == 0) ? w2 : ((r == 1) ? z2 : -y2);
        const double l2 = (r == 0) ? -z2 : ((r == 1) ? w2 : x2);
        const double l3 = (r == 0) ? y2 : ((r == 1) ? -x2 : w2);
        // column c of D = (I - q~ q~^T) / |q|
        const double tc = (c == 0) ? t0 : ((c == 1) ? t1 : ((c == 2) ? t2 : t3));
        const double d0 = ((c == 0 ? 1.0 : 0.0) - t0 * tc) / nrm;
        const double d1 = ((c == 1 ? 1.0 : 0.0) - t1 * tc) / nrm;
        const double d2 = ((c == 2 ? 1.0 : 0.0) - t2 * tc) / nrm;
        const double d3 = ((c == 3 ? 1.0 : 0.0) - t3 * tc) / nrm;
        double acc = l0 * d0;
        acc += l1 * d1;
        acc += l2 * d2;
        acc += l3 * d3;
        Nout[(qs + c) * nv + vs + r] = acc;
      }
    }
  }
  __syncthreads();
}

// v = N (qa - qb) / dt  (TO.cc:187-190); threads r < nv
IDTO_DEV double velocity_row(const DevModel& M, const double* N, const double* qa, const double* qb, double dt, int r) {
  double acc = N[r] * (qa[0] - qb[0]);
  for (int c = 1; c < M.nq; ++c) acc += N[c * M.nv + r] * (qa[c] - qb[c]);
  return acc / dt;
}
IDTO_DEV void velocity_block(const DevModel& M, const double* N, const double* qa, const double* qb, double dt,
                             double* vout, int tid, int nthreads) {
  for (int r = tid; r < M.nv; r += nthreads) vout[r] = velocity_row(M, N, qa, qb, dt, r);
}

// sum_l A[l] * B[l], l ascending, products and sums rounded separately (no FMA): THE inner product
// of the Hessian assembly (columns of nv entries, 16-byte aligned, column stride a multiple of 2)
IDTO_DEV double asm_dot(const double* A, const double* B, int nv) {
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
}
// U independent asm_dot chains in lockstep (each one the same operations in the same order as
// asm_dot: same bits)
template <int U>
IDTO_DEV void asm_dot_n(const double* const (&A)[U], const double* const (&B)[U], int nv, double (&acc)[U]) {
  const int np = nv >> 1;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const double2 a = reinterpret_cast<const double2*>(A[u])[0], b = reinterpret_cast<const double2*>(B[u])[0];
    acc[u] = a.x * b.x;
    if (nv > 1) acc[u] = acc[u] + a.y * b.y;
  }
#pragma unroll 2
  for (int m = 1; m < np; ++m) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const double2 a = reinterpret_cast<const double2*>(A[u])[m], b = reinterpret_cast<const double2*>(B[u])[m];
      acc[u] = acc[u] + a.x * b.x;
      acc[u] = acc[u] + a.y * b.y;
    }
  }
  if ((nv & 1) && nv > 1) {
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] = acc[u] + A[u][nv - 1] * B[u][nv - 1];
  }
}
// TB x TB asm_dot values acc[ur][uc] = asm_dot(A[ur], B[uc]) from 2 TB operand reads per step instead
// of 2 TB^2: the products are bound by LDS bandwidth (two 16-byte reads per two multiply-adds), a
// register tile reads each operand once for TB results.  Same operations per entry: same bits.
template <int TB>
IDTO_DEV void asm_dot_tile(const double* const (&A)[TB], const double* const (&B)[TB], int nv, double (&acc)[TB][TB]) {
  const int np = nv >> 1;
  double2 a[TB], b[TB];
#pragma unroll
  for (int u = 0; u < TB; ++u) {
    a[u] = reinterpret_cast<const double2*>(A[u])[0];
    b[u] = reinterpret_cast<const double2*>(B[u])[0];
  }
#pragma unroll
  for (int ur = 0; ur < TB; ++ur)
#pragma unroll
    for (int uc = 0; uc < TB; ++uc) {
      acc[ur][uc] = a[ur].x * b[uc].x;
      if (nv > 1) acc[ur][uc] = acc[ur][uc] + a[ur].y * b[uc].y;
    }
  // (the operands of step m + 1 are read while step m is being accumulated: one wavefront per SIMD, nothing else
  // covers the LDS round trip)
  double2 an[TB], bn[TB];
  if (np > 1) {
#pragma unroll
    for (int u = 0; u < TB; ++u) {
      an[u] = reinterpret_cast<const double2*>(A[u])[1];
      bn[u] = reinterpret_cast<const double2*>(B[u])[1];
    }
  }
  for (int m = 1; m < np; ++m) {
#pragma unroll
    for (int u = 0; u < TB; ++u) { a[u] = an[u]; b[u] = bn[u]; }
    const int mn = (m + 1 < np) ? m + 1 : m;
#pragma unroll
    for (int u = 0; u < TB; ++u) {
      an[u] = reinterpret_cast<const double2*>(A[u])[mn];
      bn[u] = reinterpret_cast<const double2*>(B[u])[mn];
    }
#pragma unroll
    for (int ur = 0; ur < TB; ++ur)
#pragma unroll
      for (int uc = 0; uc < TB; ++uc) {
        acc[ur][uc] = acc[ur][uc] + a[ur].x * b[uc].x;
        acc[ur][uc] = acc[ur][uc] + a[ur].y * b[uc].y;
      }
  }
  if ((nv & 1) && nv > 1) {
#pragma unroll
    for (int ur = 0; ur < TB; ++ur)
#pragma unroll
      for (int uc = 0; uc < TB; ++uc) acc[ur][uc] = acc[ur][uc] + A[ur][nv - 1] * B[uc][nv - 1];
  }
}

// Per tau-index k, the products of the Gauss-Newton assembly that involve record k ONLY
// (TO.cc:1127-1153, :1064-1068, with R' = 2 dt R diagonal), formed by the finite-difference
// workgroup that has the record in LDS (fd_body, `terms` != nullptr):
//   CP = P_k^T R' P_k -> C_{k+1}   CT = T_k^T R' T_k -> C_k   CM = M_k^T R' M_k -> C_{k-1}   (lower triangles)
//   BPT = P_k^T R' T_k -> B_{k+1}  BTM = T_k^T R' M_k -> B_k  APM = P_k^T R' M_k -> A_{k+1}
//   gP = P_k^T R' tau_k -> g_{k+1} gT = T_k^T R' tau_k -> g_k gM = M_k^T R' tau_k -> g_{k-1}
// assemble_terms_kernel adds them in the reference's order.  Each entry is the very expression
// assemble_diag_kernel evaluates (asm_dot on the weighted column), so both paths give the same bits.
__host__ __device__ inline int asm_terms_stride(int nq) { return 6 * nq * nq + 3 * nq + ((3 * nq) & 1); }

// ---------------------------------------------------------------------------
// fd_kernel: block <-> tau index k.  Produces slab_k = [dtau_k/dq_{k-1} |
// dtau_k/dq_k | dtau_k/dq_{k+1} | tau_k] plus v_{k+1}, a_k, N+_{k+1}.
// mode 0: tau only (one evaluation); mode 1: forward differences (TO.cc:426-563).
// Dynamic LDS layout (doubles): see the carve-up below.
// the instantiated tree shapes of id_fast.h (DevModel::fast_shape; 0 = any model: id_eval<MAXC>)
template <int SHAPE> struct FastShape { static constexpr int MAXC = 0, NP = 1, CJ = -1, J0 = 0, K0 = 0, W2 = -1; };
template <> struct FastShape<1> { static constexpr int MAXC = 2, NP = 1, CJ = -1, J0 = IDTO_JOINT_REVOLUTE, K0 = PK_WORLD, W2 = -1; };             // acrobot
template <> struct FastShape<2> { static constexpr int MAXC = 3, NP = 1, CJ = -1, J0 = IDTO_JOINT_PLANAR, K0 = PK_WORLD, W2 = -1; };               // hopper
template <> struct FastShape<3> { static constexpr int MAXC = 3, NP = 4, CJ = IDTO_JOINT_FLOATING, J0 = IDTO_JOINT_REVOLUTE, K0 = PK_COMMON, W2 = -1; };  // mini_cheetah
template <> struct FastShape<4> { static constexpr int MAXC = 4, NP = 4, CJ = IDTO_JOINT_FLOATING, J0 = IDTO_JOINT_REVOLUTE, K0 = PK_WORLD, W2 = -1; };   // allegro_hand + ball
template <> struct FastShape<5> { static constexpr int MAXC = 3, NP = 1, CJ = -1, J0 = IDTO_JOINT_REVOLUTE, K0 = PK_WORLD, W2 = 2; };              // spinner: two-link finger + the spinner, off the world

template <int MAXC, int SHAPE = 0>
IDTO_DEV void fd_body(const DevModel& M, const DevContact& cp, const DevProblem& P, const double* __restrict__ q,
                      double* __restrict__ slab, int slab_stride, double* __restrict__ v_out,
                      double* __restrict__ a_out, double* __restrict__ nplus_out, const int k, int mode,
                      int stop_after, int echunk, double* __restrict__ terms) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x, nt = blockDim.x;
  if (stop_after == 10) return;   // (profiling aid: the launch alone)
#ifdef IDTO_FD_STAMPS
  long long st_arr[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long* idto_fd_st = (tid == 0 || tid == IDTO_FD_STAMP_TID) ? st_arr : nullptr;
#endif
  FD_STAMP(0);
  const int nq = M.nq, nv = M.nv, K = M.npaths;
  const int bsz = nv * nq;
  const double dt = P.dt;

  // mode 0: tau only; 1: forward differences (+ nv mass-matrix columns for dtau_k/dq_{k-1});
  // 2 / 3: central differences of 2nd / 4th order (reference TO.cc:565-885): every partial,
  // including dtau_k/dq_{k-1}, from evaluations at q_t[i] + m*dq, m in {+1,-1} / {+1,-1,+2,-2};
  // evaluation index e = 1 + ((g * NM + mi) * nq + i), g = 0: w.r.t. q_{k+1}, 1: q_k, 2: q_{k-1}
  const bool central = mode >= 2;
  const int NM = (mode == 2) ? 2 : ((mode == 3) ? 4 : 1);
  const int nP = (mode >= 1) ? NM * nq : 0;
  const int nT = (mode >= 1) ? NM * nq : 0;
  const int nM = (mode == 1) ? nv : (central ? NM * nq : 0);
  const int E = 1 + nP + nT + nM;

  double* qm1 = lds;            // q_{k-1}
  double* q0 = qm1 + nq;        // q_k
  double* q1 = q0 + nq;         // q_{k+1}
  double* N0 = q1 + nq;         // N+_k
  double* N1 = N0 + bsz;        // N+_{k+1}
  double* v0 = N1 + bsz;        // v_k
  double* v1 = v0 + nv;         // v_{k+1}
  double* a0 = v1 + nv;         // a_k
  double* edq = a0 + nv;        // [E] perturbation of each evaluation, then [E] dq / dt and [E] dq / dt^2
  // the evaluation inputs are built `echunk` evaluations at a time (one pass unless the
  // central-difference evaluation set of a large model would not fit in LDS)
  const int EC = (echunk < E) ? echunk : E;
  double* edv = edq + E;
  double* eda = edv + E;
  double* etau = eda + E;       // [E][nv]
  double* eq = etau + E * nv;   // [EC][nq]
  double* ev = eq + EC * nq;    // [EC][nv]
  double* ea = ev + EC * nv;    // [EC][nv]
  double* edump = ea + EC * nv; // [nv] write-only dump row for surplus lanes
  double* mblob = edump + nv;    // [M.blob_n] the model tables
  // (a kernel of a fast shape needs the gathered records and three int tables only: DevModel::fast_lo / fast_n)
  const int blob_lo = (SHAPE != 0) ? M.fast_lo : 0, blob_n = (SHAPE != 0) ? M.fast_n : M.blob_n;
  int* colinfo = reinterpret_cast<int*>(mblob + blob_n + (blob_n & 1));  // [nq] non-zero rows of N+ column c
  // (terms != nullptr) the record and its weighted copy for the assembly products: 6 blocks of nq
  // columns, column stride nvp (16-byte aligned columns), + tau_k R' and the diagonal of R'
  const int nvp = (nv + 1) & ~1, psz = nvp * nq;
  // (16-byte aligned: asm_dot reads double2; 8 bytes off, the products below took twice as long)
  double* rec = reinterpret_cast<double*>(colinfo + nq + (nq & 1));   // [P | T | M | P R' | T R' | M R'] then diag R'
  rec += (rec - lds) & 1;

  // N+_k and N+_{k+1} (TO.cc:1633-1647).  The model's table holds the constant entries (NaN where a quaternion block
  // goes); the 3 x 4 block of a floating joint is formed here, entry by entry, by twelve lanes of wavefront 0 (q_k) and of
  // wavefront 1 (q_{k+1}) straight from the global q - the expressions of nplus_pair, hence the same bits - so that N+
  // is complete behind the ONE barrier that also covers the loads.  (More than four floating joints: nplus_pair.)
  // Their quaternions are the first loads the kernel issues: the longest chain of the prologue hangs on them.
  const bool sparse = M.nfloat == 0 || (M.nfloat > 0 && nt >= 128);
  const int ql = tid & 63, qcfg = tid >> 6;   // wavefront 0: q_k, wavefront 1: q_{k+1}
  const bool quat_lane = sparse && M.nfloat > 0 && qcfg < 2 && ql >= 48 && ql < 60;
  double qf0[4];   // the quaternion of the first floating joint (read by every thread: no merge under an exec mask, no wait here)
  {
    const double* qq = q + (size_t)(k + (qcfg & 1)) * nq + (M.nfloat > 0 ? M.float_qs[0] : 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) qf0[i] = qq[i];
  }
  // stage the model into LDS and use that copy from here on.  Every global read of the prologue is issued before the
  // first of them is used: a loop `lds[i] = global[i]` waits for each of its loads in turn, and after a kernel
  // boundary none of them is an L2 hit - ten such round trips in a row were 7k of the kernel's 48k cycles.
  double* wr = rec + 6 * psz;      // [nv] diagonal of R' (fetched now: an HBM round trip off the tail's critical path)
  constexpr int BU = 8;            // model words per thread in the batched part (the rest, if any, in a loop)
  double breg[BU];
#pragma unroll
  for (int u = 0; u < BU; ++u) {
    const int i = tid + u * nt;
    breg[u] = (i < blob_n) ? M.blob[blob_lo + i] : 0.0;
  }
  const bool qrow = tid < nq;
  const double g_qm1 = (qrow && k > 0) ? q[(k - 1) * nq + tid] : 0.0;
  const double g_q0 = qrow ? q[k * nq + tid] : 0.0, g_q1 = qrow ? q[(k + 1) * nq + tid] : 0.0;
  const double g_wr = (terms && mode != 0 && tid < nv) ? P.R[tid * nv + tid] : 0.0;
  const double g_nc0 = (sparse && tid < bsz) ? M.nplus_const[tid] : 0.0;
  const double g_nc1 = (sparse && tid + nt < bsz) ? M.nplus_const[tid + nt] : 0.0;
  const int g_ci = (sparse && qrow) ? M.colinfo[tid] : 0;
  const int rowinfo_r = (sparse && tid < nv) ? M.rowinfo[tid] : 0;
  // (the quaternion pinned here: its loads are above this line - the compiler otherwise sinks them to their use, behind
  // the staging stores - and, issued among the first, they are back before the others)
  asm volatile("" : "+v"(qf0[0]), "+v"(qf0[1]), "+v"(qf0[2]), "+v"(qf0[3]));
#pragma unroll
  for (int u = 0; u < BU; ++u) {
    const int i = tid + u * nt;
    if (i < blob_n) mblob[i] = breg[u];
  }
  for (int i = tid + BU * nt; i < blob_n; i += nt) mblob[i] = M.blob[blob_lo + i];
  if (qrow) { qm1[tid] = g_qm1; q0[tid] = g_q0; q1[tid] = g_q1; }
  for (int i = tid + nt; i < nq; i += nt) {
    qm1[i] = (k > 0) ? q[(k - 1) * nq + i] : 0.0;
    q0[i] = q[k * nq + i];
    q1[i] = q[(k + 1) * nq + i];
  }
  if (terms && mode != 0) {
    if (tid < nv) wr[tid] = g_wr;
    for (int l = tid + nt; l < nv; l += nt) wr[l] = P.R[l * nv + l];
  }
  if (sparse) {
    if (tid < bsz && g_nc0 == g_nc0) { N0[tid] = g_nc0; N1[tid] = g_nc0; }
    if (tid + nt < bsz && g_nc1 == g_nc1) { N0[tid + nt] = g_nc1; N1[tid + nt] = g_nc1; }
    for (int i = tid + 2 * nt; i < bsz; i += nt) {
      const double cst = M.nplus_const[i];
      if (cst == cst) { N0[i] = cst; N1[i] = cst; }
    }
    if (qrow) colinfo[tid] = g_ci;
    for (int i = tid + nt; i < nq; i += nt) colinfo[i] = M.colinfo[i];
    if (quat_lane) {
      const int e = ql - 48, r = e >> 2, c = e & 3;
      double* Nout = qcfg ? N1 : N0;
      for (int f = 0; f < M.nfloat; ++f) {
        const int qs = M.float_qs[f], vs = M.float_vs[f];
        double qq[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) qq[i] = (f == 0) ? qf0[i] : q[(size_t)(k + qcfg) * nq + qs + i];
        const double nrm = __builtin_sqrt(((qq[0] * qq[0] + qq[1] * qq[1]) + qq[2] * qq[2]) + qq[3] * qq[3]);
        const double t0 = qq[0] / nrm, t1 = qq[1] / nrm, t2 = qq[2] / nrm, t3 = qq[3] / nrm;
        const double w2 = 2.0 * t0, x2 = 2.0 * t1, y2 = 2.0 * t2, z2 = 2.0 * t3;
        // row r of LT = L(2 q~)^T
        const double l0 = (r == 0) ? -x2 : ((r == 1) ? -y2 : -z2);
        const double l1 = (r == 0) ? w2 : ((r == 1) ? z2 : -y2);
        const double l2 = (r == 0) ? -z2 : ((r == 1) ? w2 : x2);
        const double l3 = (r == 0) ? y2 : ((r == 1) ? -x2 : w2);
        // column c of D = (I - q~ q~^T) / |q|
        const double tc = (c == 0) ? t0 : ((c == 1) ? t1 : ((c == 2) ? t2 : t3));
        const double d0 = ((c == 0 ? 1.0 : 0.0) - t0 * tc) / nrm;
        const double d1 = ((c == 1 ? 1.0 : 0.0) - t1 * tc) / nrm;
        const double d2 = ((c == 2 ? 1.0 : 0.0) - t2 * tc) / nrm;
        const double d3 = ((c == 3 ? 1.0 : 0.0) - t3 * tc) / nrm;
        double acc = l0 * d0;
        acc += l1 * d1;
        acc += l2 * d2;
        acc += l3 * d3;
        Nout[(qs + c) * nv + vs + r] = acc;
      }
    }
  } else {
    for (int i = tid; i < bsz; i += nt) { N0[i] = 0.0; N1[i] = 0.0; }   // (nplus_pair fills the non-zeros behind this barrier)
  }
  FD_STAMP(1);
  __syncthreads();
  FD_STAMP(2);
  if (stop_after == 8) return;    // ... after the loads
  const DevModel Ml = rebase_model(M, mblob - blob_lo);
  if (!sparse) {
    // N+ is sparse: column c has its non-zeros in rows [j0, j0 + cnt) (cnt = 3 for the quaternion
    // columns, 1 otherwise); packed j0 | cnt << 16
    for (int b = tid; b < Ml.nb; b += nt) {
      const int qs = Ml.qstart[b], vs = Ml.vstart[b], jt = Ml.jtype[b];
      if (jt == IDTO_JOINT_REVOLUTE || jt == IDTO_JOINT_PRISMATIC) colinfo[qs] = vs | 1 << 16;
      else if (jt == IDTO_JOINT_PLANAR) { for (int kq = 0; kq < 3; ++kq) colinfo[qs + kq] = (vs + kq) | 1 << 16; }
      else {
        for (int kq = 0; kq < 4; ++kq) colinfo[qs + kq] = vs | 3 << 16;
        for (int kq = 0; kq < 3; ++kq) colinfo[qs + 4 + kq] = (vs + 3 + kq) | 1 << 16;
      }
    }
    nplus_pair(Ml, q0, N0, q1, N1, tid, nt, true);
  }
  if (stop_after == 9) return;    // ... after N+
  // v_k, v_{k+1} and a_k = (v_{k+1} - v_k) / dt of row r by ONE thread: no barrier between velocities and acceleration
  // (the expressions are velocity_block's and the reference's, TO.cc:187-201; with the table of the rows' non-zero
  // ranges the terms N+(r, c) (q_c - q'_c) with N+(r, c) == 0 are left out: they are zeros of either sign, so that
  // the sum is the same number and can differ from the full one in the sign of a zero result only)
  for (int r = tid; r < nv; r += nt) {
    double vr0, vr1;
    if (sparse) {
      const int ri = (r == tid) ? rowinfo_r : M.rowinfo[r];
      const int c0 = ri & 0xffff, c1 = c0 + (ri >> 16);
      double acc0 = N0[c0 * nv + r] * (q0[c0] - qm1[c0]), acc1 = N1[c0 * nv + r] * (q1[c0] - q0[c0]);
      for (int c = c0 + 1; c < c1; ++c) {
        acc0 += N0[c * nv + r] * (q0[c] - qm1[c]);
        acc1 += N1[c * nv + r] * (q1[c] - q0[c]);
      }
      vr0 = (k > 0) ? acc0 / dt : P.v_init[r];
      vr1 = acc1 / dt;
    } else {
      vr0 = (k > 0) ? velocity_row(Ml, N0, q0, qm1, dt, r) : P.v_init[r];
      vr1 = velocity_row(Ml, N1, q1, q0, dt, r);
    }
    v0[r] = vr0;
    v1[r] = vr1;
    a0[r] = (vr1 - vr0) / dt;
  }
  // the perturbation of every evaluation (TO.cc:501-521; needs q only: formed here, a barrier earlier)
  const double eps = 1.4901161193847656e-08;  // sqrt(DBL_EPSILON) = 2^-26
  // (by the threads from the second wavefront on: the first ones have the rows of v and a, two divisions deep as well)
  for (int e = (nt > 64) ? (tid + nt - 64) % nt : tid; e < E; e += nt) {
    double dq = 1.0;
    if (e >= 1 && (e < 1 + nP + nT || central)) {
      const int i = (e - 1) % nq;
      const double qi = (e < 1 + nP) ? q1[i] : ((e < 1 + nP + nT) ? q0[i] : qm1[i]);
      dq = eps * __builtin_fmax(1.0, __builtin_fabs(qi));
      const double temp = qi + dq;
      dq = temp - qi;
    }
    edq[e] = dq;
    const double dv = dq / dt;   // (once per evaluation: an f64 division is ~30 dependent instructions, and the
    edv[e] = dv;                 // input loops below would repeat both for every velocity component)
    eda[e] = dv / dt;
  }
  __syncthreads();

  // trajectory outputs, by the threads t0, t0 + ts, ...
  auto trajectory_outputs = [&](int t0, int ts) {
    for (int r = t0; r < nv; r += ts) {
      v_out[(k + 1) * nv + r] = v1[r];
      a_out[k * nv + r] = a0[r];
      if (k == 0) v_out[r] = v0[r];
    }
    for (int i = t0; i < bsz; i += ts) {
      nplus_out[(size_t)(k + 1) * bsz + i] = N1[i];
      if (k == 0) nplus_out[i] = N0[i];
    }
  };
  // (a fast shape with forward differences: the last wavefront writes them after its evaluations - the cheap
  // mass-matrix columns - while the others are still in their contact pairs)
  const bool late_outputs = (SHAPE != 0) && !central && stop_after != 1;
  if (!late_outputs) trajectory_outputs(tid, nt);

  FD_STAMP(3);
  if (stop_after == 1) return;  // (profiling aid: phase timing by truncation) after N+, v, a
  // ---- evaluation inputs (TO.cc:501-521): the perturbations were formed with N+ / v above
  // (a fast shape with forward differences: every lane forms the inputs of its own evaluation, id_fast.h InFwd)
  const bool own_inputs = (SHAPE != 0) && !central;
  for (int c0 = 0; c0 < E; c0 += (own_inputs ? E : EC)) {
  const int ce = own_inputs ? E : ((E - c0 < EC) ? E - c0 : EC);  // evaluations [c0, c0 + ce) in this pass
  if (!own_inputs) {
  for (int idx = tid; idx < ce * nq; idx += nt) {
    const int el = idx / nq, c = idx - el * nq, e = c0 + el;
    double val = q1[c];
    if (e >= 1 && e < 1 + nP && c == (e - 1) % nq) {
      if (!central) {
        val = q1[c] + edq[e];
      } else {  // qi + mult * dq (TO.cc:766)
        const int mi = (e - 1) / nq;
        const double mult = (mi == 0) ? 1.0 : ((mi == 1) ? -1.0 : ((mi == 2) ? 2.0 : -2.0));
        val = q1[c] + mult * edq[e];
      }
    }
    eq[idx] = val;
  }
  for (int idx = tid; idx < ce * nv; idx += nt) {
    const int el = idx / nv, j = idx - el * nv, e = c0 + el;
    double vv = v1[j], aa = a0[j];
    if (central && e >= 1) {
      const int g = (e - 1) / (NM * nq), mi = ((e - 1) / nq) % NM, i = (e - 1) % nq;
      const double mult = (mi == 0) ? 1.0 : ((mi == 1) ? -1.0 : ((mi == 2) ? 2.0 : -2.0));
      const double dv = edv[e], da = eda[e];
      const double n1 = N1[i * nv + j], n0 = N0[i * nv + j];
      if (g == 0) {         // tau_k(q_{k+1} + m dq): TO.cc:763-787
        vv = v1[j] + (mult * dv) * n1;
        aa = a0[j] + (mult * da) * n1;
      } else if (g == 1) {  // tau_k under q_k + m dq: :788-814
        vv = v1[j] - (mult * dv) * n1;
        aa = a0[j] - (mult * da) * (n1 + n0);
      } else {              // tau_k under q_{k-1} + m dq: :815-839
        aa = a0[j] + (mult * da) * n0;
      }
    } else if (e >= 1 && e < 1 + nP) {
      const int i = e - 1;
      const double dv = edv[e], da = eda[e];
      const double n1 = N1[i * nv + j];
      vv = v1[j] + dv * n1;
      aa = a0[j] + da * n1;
    } else if (e >= 1 + nP && e < 1 + nP + nT) {
      const int i = e - 1 - nP;
      const double dv = edv[e], da = eda[e];
      const double n1 = N1[i * nv + j], n0 = N0[i * nv + j];
      vv = v1[j] - dv * n1;
      aa = a0[j] - da * (n1 + n0);
    } else if (e >= 1 + nP + nT) {
      vv = 0.0;
      aa = (j == e - 1 - nP - nT) ? 1.0 : 0.0;
    }
    ev[idx] = vv;
    ea[idx] = aa;
  }
  __syncthreads();
  }

  if (stop_after == 2) return;  // after the evaluation inputs
  // ---- the evaluations: lane = (evaluation, path)
  // Surplus groups (e >= E) re-run evaluation 0 into a dump row so that every lane
  // of a wavefront takes part in the butterfly sums inside id_eval.
  const int groups = nt / K;
  if constexpr (SHAPE != 0) {
    // Straight-line evaluation (id_fast.h).  The lanes of a surplus group stay idle (a group's butterfly
    // partners are its own lanes); evaluations beyond the first round go to the LAST groups: with forward
    // differences those hold the cheap mass-matrix columns, as do the evaluations left over (allegro:
    // 69 evaluations on 64 groups - the wavefront of the contact-free columns takes the five extra ones).
    using FS = FastShape<SHAPE>;
    FastTab FT;
    FT.body = Ml.f_body; FT.cbody = Ml.f_cbody; FT.pairs = Ml.f_pairs; FT.seg = Ml.f_seg; FT.maxpp = Ml.f_maxpp;
    const int grp = tid / FS::NP, path = tid % FS::NP;
    for (int e0 = 0; e0 < ce; e0 += groups) {
      const int el = (e0 == 0) ? grp : e0 + (groups - 1 - grp);
      if (el < ce) {
        const bool full = central || c0 + el < 1 + nP + nT;
        if (own_inputs) {
          InFwd in;
          in.q1 = q1; in.v1 = v1; in.a0 = a0; in.N1 = N1; in.N0 = N0; in.nv = nv;
          in.kind = (el == 0) ? 0 : ((el < 1 + nP) ? 1 : ((el < 1 + nP + nT) ? 2 : 3));
          in.col = (el == 0) ? 0 : ((el < 1 + nP) ? el - 1 : ((el < 1 + nP + nT) ? el - 1 - nP : el - 1 - nP - nT));
          in.dq = edq[el];
          in.sdv = (in.kind == 2) ? -edv[el] : ((in.kind == 1) ? edv[el] : 0.0);
          in.sda = (in.kind == 2) ? -eda[el] : ((in.kind == 1) ? eda[el] : 0.0);
          in.keep = (in.kind == 3) ? 0ull : ~0ull;
          in.keep0 = (in.kind == 2) ? ~0ull : 0ull;
          id_eval_fast<FS::MAXC, FS::NP, FS::CJ, FS::J0, FS::K0, FS::W2>(FT, Ml.gravity, cp, path, full, in, etau + el * nv
#ifdef IDTO_FD_STAMPS
                                                                 , (e0 == 0) ? idto_fd_st : nullptr
#endif
          );
        } else {
          InLds in;
          in.q = eq + el * nq; in.v = ev + el * nv; in.a = ea + el * nv;
          id_eval_fast<FS::MAXC, FS::NP, FS::CJ, FS::J0, FS::K0, FS::W2>(FT, Ml.gravity, cp, path, full, in, etau + (c0 + el) * nv);
        }
      }
    }
    if (late_outputs && tid >= nt - 64) trajectory_outputs(tid - (nt - 64), 64);
  } else {
  for (int e0 = 0; e0 < ce; e0 += groups) {
    const int el = e0 + tid / K;
    const int path = tid % K;
    const int ee = (el < ce) ? el : 0;
    const bool full = central || c0 + ee < 1 + nP + nT;
    double* tau_dst = (el < ce) ? etau + (c0 + ee) * nv : edump;
    id_eval<MAXC>(Ml, cp, path, full, eq + ee * nq, ev + ee * nv, ea + ee * nv, tau_dst);
  }
  }
  __syncthreads();
  }

  FD_STAMP(12);
  if (stop_after == 3) return;  // after the inverse-dynamics evaluations
  // ---- outputs
  double* sl = slab + (size_t)k * slab_stride;
  double* Mk = sl;
  double* Tk = sl + bsz;
  double* Pk = sl + 2 * bsz;
  double* tauk = sl + 3 * bsz;
  for (int r = tid; r < nv; r += nt) tauk[r] = etau[r];
  if (mode == 1) {
    // Two entries of each of the three blocks per thread and pass, every LDS read of the pass before the first division:
    // one wavefront per SIMD has nothing else to cover the reads' and the divisions' latencies with.
    //   dtau_k/dq_{k+1} (TO.cc:531), dtau_k/dq_k (:539), dtau_k/dq_{k-1} = (1/dt^2) M(q_{k+1}) N+_k (:556-561): the sum over j
    //   in the reference's order; the terms outside [j0, j0 + cnt) are exact zeros
    const double sc = 1 / dt / dt;
    const double* Mcols = etau + (1 + nP + nT) * nv;  // column j = tau of mass evaluation j
    const double fillM = (k == 0) ? __builtin_nan("") : 0.0;
    constexpr int RU = 2;
    for (int base = tid; base < bsz; base += RU * nt) {
      int ii[RU], rr[RU], j0[RU], cnt[RU];
      bool ok[RU];
      double tp[RU], tt[RU], t0[RU], dqp[RU], dqt[RU], w[RU], mc[RU][3], n0[RU][3];
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        const int idx = base + u * nt;
        ok[u] = idx < bsz;
        const int id = ok[u] ? idx : 0;
        ii[u] = id / nv; rr[u] = id - ii[u] * nv;
        const int ci = colinfo[ii[u]];
        j0[u] = ci & 0xffff; cnt[u] = ci >> 16;
        tp[u] = etau[(1 + ii[u]) * nv + rr[u]]; tt[u] = etau[(1 + nP + ii[u]) * nv + rr[u]]; t0[u] = etau[rr[u]];
        dqp[u] = edq[1 + ii[u]]; dqt[u] = edq[1 + nP + ii[u]];
        w[u] = terms ? wr[rr[u]] : 0.0;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const int j = (t < cnt[u]) ? j0[u] + t : j0[u];
          mc[u][t] = Mcols[j * nv + rr[u]];
          n0[u][t] = N0[ii[u] * nv + j];
        }
      }
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        if (!ok[u]) continue;
        const int idx = base + u * nt, i = ii[u], r = rr[u];
        const double pv = (tp[u] - t0[u]) / dqp[u];
        const double tv = (k >= 1) ? (tt[u] - t0[u]) / dqt[u] : 0.0;
        double acc = (sc * mc[u][0]) * n0[u][0];
        if (cnt[u] > 1) acc += (sc * mc[u][1]) * n0[u][1];
        if (cnt[u] > 2) acc += (sc * mc[u][2]) * n0[u][2];
        const double mv = (k >= 2) ? acc : fillM;
        Pk[idx] = pv;
        Tk[idx] = tv;
        Mk[idx] = mv;
        if (terms) {   // ... and (A^T W)(r, l) = A(l, r) w_l, as assemble_diag_kernel forms it, by the thread that holds A(l, r)
          const double mr = (k >= 2) ? acc : 0.0;   // (never used by the assembly for k < 2)
          rec[i * nvp + r] = pv; rec[psz + i * nvp + r] = tv; rec[2 * psz + i * nvp + r] = mr;
          rec[3 * psz + i * nvp + r] = pv * w[u]; rec[4 * psz + i * nvp + r] = tv * w[u]; rec[5 * psz + i * nvp + r] = mr * w[u];
          if (r == nv - 1 && nvp > nv) {
            rec[3 * psz + i * nvp + nv] = 0.0; rec[4 * psz + i * nvp + nv] = 0.0; rec[5 * psz + i * nvp + nv] = 0.0;
          }
        }
      }
    }
  } else if (central) {
    // (tau(+) - tau(-)) / (2 dq), or the five-point formula, in the reference's expression order
    for (int idx = tid; idx < 3 * bsz; idx += nt) {
      const int g = idx / bsz, rem = idx - g * bsz, i = rem / nv, r = rem - i * nv;
      const int e0 = 1 + (g * NM) * nq + i;  // multiplier 0 (+1); +nq per multiplier index
      const double dq = edq[e0];
      const double tp = etau[e0 * nv + r], tm = etau[(e0 + nq) * nv + r];
      double d;
      if (mode == 2) {
        d = 0.5 * (tp - tm) / dq;
      } else {
        const double tpp = etau[(e0 + 2 * nq) * nv + r], tmm = etau[(e0 + 3 * nq) * nv + r];
        d = 2.0 / 3.0 * (tp - tm) / dq - 1.0 / 12.0 * (tpp - tmm) / dq;
      }
      const double val = (g == 0) ? d : ((g == 1) ? ((k >= 1) ? d : 0.0) : ((k >= 2) ? d : ((k == 0) ? __builtin_nan("") : 0.0)));
      if (g == 0) Pk[rem] = val;
      else if (g == 1) Tk[rem] = val;
      else Mk[rem] = val;
      if (terms) {
        const double rv = (g == 2 && k < 2) ? 0.0 : val;
        rec[g * psz + i * nvp + r] = rv;
        rec[(3 + g) * psz + i * nvp + r] = rv * wr[r];
        if (r == nv - 1 && nvp > nv) rec[(3 + g) * psz + i * nvp + nv] = 0.0;
      }
    }
  }
  if (!terms || mode == 0 || stop_after == 4) return;
  // ---- the assembly products of this record (see asm_terms_stride); the weighted copy was formed with the record
  FD_STAMP(13);
  __syncthreads();
  if (stop_after == 5) return;
  const int qq = nq * nq, ts = asm_terms_stride(nq);
  // (each product goes straight to its place in HBM: stores do not stall the lane, and a staging pass through LDS
  // behind one more barrier was 2.9k of the kernel's 48k cycles)
  double* stage = terms + (size_t)k * ts;
  // 3 x 3 register tiles: the lower-triangle tiles of CP, CT, CM, then all tiles of
  // BPT = (P R')^T T, BTM = (T R')^T M, APM = (P R')^T M  (one round of 231 tiles at nq = 19)
  constexpr int TB = 3;
  const int nb = (nq + TB - 1) / TB, ntt = nb * (nb + 1) / 2, nbb = nb * nb, ntiles = 3 * ntt + 3 * nbb;
  const float inb = 1.0f / (float)nb;
  for (int tile = tid; tile < ntiles; tile += nt) {
    if (stop_after == 6 && tile >= 3 * ntt) break;
    int xa, sb, tr, tc, ob;
    const bool lower = tile < 3 * ntt;
    if (lower) {
      const int which = (tile >= ntt) + (tile >= 2 * ntt);
      int rem = tile - which * ntt;
      tc = 0;
      while (rem >= nb - tc) { rem -= nb - tc; ++tc; }
      tr = tc + rem;
      xa = which; sb = which; ob = which * qq;
    } else {
      const int it = tile - 3 * ntt, which = (it >= nbb) + (it >= 2 * nbb), e = it - which * nbb;
      tc = (int)(((float)e + 0.5f) * inb);   // e / nb (exact: e < 2^12)
      tr = e - tc * nb;
      xa = (which == 1) ? 1 : 0; sb = (which == 0) ? 1 : 2; ob = (3 + which) * qq;
    }
    const double* A[TB];
    const double* B[TB];
#pragma unroll
    for (int u = 0; u < TB; ++u) {   // (rows / columns past the edge: recompute the last one, not stored)
      const int r = tr * TB + u, c = tc * TB + u;
      A[u] = rec + (3 + xa) * psz + (r < nq ? r : nq - 1) * nvp;
      B[u] = rec + sb * psz + (c < nq ? c : nq - 1) * nvp;
    }
    double acc[TB][TB];
    asm_dot_tile<TB>(A, B, nv, acc);
#pragma unroll
    for (int ur = 0; ur < TB; ++ur)
#pragma unroll
      for (int uc = 0; uc < TB; ++uc) {
        const int r = tr * TB + ur, c = tc * TB + uc;
        if (r < nq && c < nq && (!lower || r >= c)) stage[ob + c * nq + r] = acc[ur][uc];
      }
  }
  if (stop_after != 7)
    for (int it = nt - 1 - tid; it < 3 * nq; it += nt) {   // gP, gT, gM: sum_r (tau_r w_r) J[r][j]  (threads without a tile first)
      const int which = (it >= nq) + (it >= 2 * nq), j = it - which * nq;
      const double* J = rec + which * psz + j * nvp;
      double acc = (etau[0] * wr[0]) * J[0];
      for (int r = 1; r < nv; ++r) acc += (etau[r] * wr[r]) * J[r];
      stage[6 * qq + which * nq + j] = acc;
    }
  FD_STAMP(14);
  FD_STAMP(15);
#ifdef IDTO_FD_STAMPS
  if (idto_fd_st && k == 1)
    for (int i = 0; i < 16; ++i) nplus_out[(size_t)(k + 1) * bsz + (tid == 0 ? 0 : 16) + i] = (double)(st_arr[i] - st_arr[0]);   // (over N+_2: a profiling build)
#endif
}

template <int MAXC, int SHAPE = 0>
__global__ void __launch_bounds__(256) fd_kernel(DevModel M, DevContact cp, DevProblem P, const double* __restrict__ q,
                          double* __restrict__ slab, int slab_stride, double* __restrict__ v_out,
                          double* __restrict__ a_out, double* __restrict__ nplus_out, int k_begin, int mode,
                          int stop_after, int echunk, size_t pstride, double* __restrict__ terms, AltSel alt) {
  const size_t o = (size_t)blockIdx.y * pstride;   // problem of the batch
  const size_t w = o + (size_t)alt_offset(alt, o);    // ... and the set of outputs (batch.h AltSel)
  fd_body<MAXC, SHAPE>(M, cp, at_problem(P, o), at_problem(q, o), at_problem(slab, w), slab_stride, at_problem(v_out, w),
                       at_problem(a_out, w), at_problem(nplus_out, w), k_begin + (int)blockIdx.x, mode, stop_after, echunk,
                       terms ? at_problem(terms, w) : nullptr);
}


}  // namespace idto_dev
