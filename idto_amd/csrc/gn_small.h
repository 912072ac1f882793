// gn_small.h — the whole Gauss-Newton step of a SMALL model in ONE workgroup of ONE launch.
//
// acrobot (2 DoF) and the spinner (3 DoF, one contact pair) at N = 40 are 280 / 400 inverse-dynamics evaluations, 82 /
// 123 unknowns: the two-launch step (fd_kernel on 40 workgroups, then penta_band_kernel with 164 assembly workgroups
// in front of its one solver workgroup) spends most of its 28 / 39 us on what lies BETWEEN workgroups - a kernel
// boundary, the cold first touch behind it, the assembly's write-through stores, their acknowledgement, the solver's
// poll and its loads (penta_band.h: "g, H assembled 5.2 us after the launch's start") - and on the CPU port's side of
// the table eight threads are faster (43.8k against 35.4k it/s for acrobot).  Here every phase of
// reference optimizer/trajectory_optimizer.cc's iteration runs in the LDS of one compute unit:
//
//   A  loads        model records, q (all N + 1 rows), the weights' diagonals                          (one round of loads)
//   B  v, a, dq     TO.cc:178-202, :501-521 - fd_body's expressions (fd_kernel.h), a thread per (t, row) / (k, evaluation)
//   C  evaluations  lane = (k, e): id_eval_fast<SHAPE> (id_fast.h) with fd_body's InFwd gather, all N (1 + 3 nq) at once
//   D  records      dtau_k/dq_{k-1,k,k+1} (TO.cc:527-561) -> the slab (API: IDTO_ARR_DTAU_*) and LDS
//   E  assembly     g, the bands (TO.cc:1021-1165): assemble_diag_body's sums (kernels.h), a thread per output entry
//   F  solve        penta_band_body (penta_band.h): the scalar band LDL^T, two wavefronts
//
// Same expressions in the same order as the kernels it stands in for, hence the same bits:
// tests/test_gpu_small.py holds every array of the step == the two-launch path (which is == the oracle).
// Serves: models of fast shape 1 / 5 (all joints revolute: N+ is the constant table, nq = nv <= 4), forward differences,
// diagonal cost weights, single-problem contexts or batches (grid.y = problem); anything else keeps the two launches.
#pragma once

#include "kernels.h"
#include "penta_band.h"

namespace idto_dev {

struct SmallArgs {
  DevModel M;
  DevContact cp;
  DevProblem P;
  const double* q;
  double* slab; int slab_stride;
  double *v, *a, *nplus;
  double *g, *HA, *HB, *HC;
  size_t pstride;
  AltSel alt;
  BandArgs B;       // the solver's view (sub-system from block row 1 on)
  int lds_small;    // where this kernel's own arrays start (doubles): behind the band solver's carve-up
  double* ts;       // option "solver_debug" 4: wall-clock stamps of the phases (100 MHz), else nullptr
  const BandStageItem* stage; int nstage;   // penta_band.h band_stage_table for this horizon, sources relative to the LDS copy of [A | B | C | g]
  // Inside the resident trust-region loop (idto_hip_tr_solve, T.state set): `q` is the TRIAL point of the iteration; behind
  // the records the workgroup evaluates the cost (cost_kernel's sums in cost_kernel's order), decides (tr_decide) and -
  // accepted - goes on to g, H and the next step; rejected, it stops there: g, H and the step are the iterate's.  One launch
  // stands in for fd_kernel, cost_kernel and the solver's launch with the gated assembly in front.
  TrDecideArgs T;
  int tau_only;        // the loop's last iteration: tau and the cost of the trial point, the decision; no partials, no step
  double* cost_out;
  // ... with enforced equality constraints (template parameter WS = 3 (nq + nu) > W): the step is the banded KKT solve of
  // kkt.h - kkt_build_kernel's bands formed in LDS from the bands, the records and tau this workgroup holds, `B` the KKT
  // context's solver view (x = z, Dst), T.dofs the constrained degrees of freedom
  int kkt_r0;          // first block row the KKT solver works on
  size_t kstride;      // the KKT context's arena stride (bytes)
  // ... and tr_iter_kernel's part in front of it all (I.state set; grid.x = 2, block 1 is tr_status_reader): scale factors,
  // g~, w, the band products, the ten sums per block row in tr_prepare_rows_body's order, the block rows in order, the
  // convergence criteria and the dogleg (tr_conv_dogleg), dq and the trial point - which then never leaves the workgroup.
  // The whole iteration is ONE launch.
  TrIterArgs I;
};
// LDS of the folded iteration (doubles); it has to fit the band solver's carve-up (the host checks)
__host__ __device__ constexpr int gn_small_fold_doubles(int N, int K) { return 26 * (N + 1) * K + 18 * (N + 1) + 32; }

// doubles of dynamic LDS behind the band solver's carve-up (gn_small_kernel's own arrays, in its order)
__host__ __device__ inline int gn_small_doubles(int N, int K, int fast_n, int KK = 0) {
  const int E = 1 + 3 * K;
  if (KK > K) return gn_small_doubles(N, K, fast_n) + 3 * (N + 1) * KK * KK + (N + 1) * KK;
  return 2 * (N + 1) * K + N * K + 3 * N * E + N * E * K + K * K + 3 * N * K * K + 5 * K + (K & 1) + fast_n + (fast_n & 1) +
         3 * (N + 1) * K * K + 3 * (N + 1) * K + 2 * K * K + (K * K & 1) + K + 2 +
         5 * K + (K & 1) + (3 * N + 2) * (K + 1) + 2 * (N + 1) + 2 + 24;   // (the trust-region loop's cost and decision)
}

// TR: the instantiation that serves the trust-region loop (T / I / tau_only are looked at); the plain step's has none of that
// code - with it in, the plain step lost 1.2 us to the register allocation alone (acrobot 20.9 -> 22.2 us).
template <int SHAPE, int W, int NT, int WS = W, bool TR = false>
__global__ void __launch_bounds__(NT) gn_small_kernel(SmallArgs S) {
  extern __shared__ double lds[];
  using FS = FastShape<SHAPE>;
  static_assert(FS::NP == 1 && FS::CJ < 0, "one path, no common body");
  constexpr int K = W / 3;                      // nq = nv
  constexpr int KK = WS / 3, NU = KK - K;       // the solver's block: nq (+ nu multiplier rows of the KKT system)
  const int tid = threadIdx.x, nt = blockDim.x;
  const size_t o = (size_t)blockIdx.y * S.pstride, w = o + (size_t)alt_offset(S.alt, o);
  if (TR && blockIdx.x == 1) {   // (folded iteration) the solver's status words over PCIe, the multiplier pivots' range
    TrIterArgs I = S.I;
    I.rows.part_ll = at_problem(I.rows.part_ll, o);
    if (I.fact_status) I.fact_status += 2 * blockIdx.y;
    if (I.rows.kx.z) I.kdinv = at_problem(I.kdinv, (size_t)blockIdx.y * I.kstride);
    tr_status_reader(I, I.rows.nblk, K, tid);
    return;
  }
  const DevProblem P = at_problem(S.P, o);
  const double* q = at_problem(S.q, o);
  double* slab = at_problem(S.slab, w);
  double* v_out = at_problem(S.v, w);
  double* a_out = at_problem(S.a, w);
  double* nplus_out = at_problem(S.nplus, w);
  double* g = at_problem(S.g, o);
  double* HA = at_problem(S.HA, o);
  double* HB = at_problem(S.HB, o);
  double* HC = at_problem(S.HC, o);
  const DevModel& M = S.M;
  const int N = P.N, nq = K, nv = K, bsz = K * K, qq = K * K;
  const int nP = nq, nT = nq, nM = nv, E = 1 + nP + nT + nM;
  const double dt = P.dt;

  // ---- carve-up (doubles), behind the band solver's
  double* qs = lds + S.lds_small;          // [(N + 1) K]
  double* vs = qs + (N + 1) * K;           // [(N + 1) K]
  double* as = vs + (N + 1) * K;           // [N K]
  double* edq = as + N * K;                // [N E], then dq / dt, dq / dt^2
  double* edv = edq + N * E;
  double* eda = edv + N * E;
  double* etau = eda + N * E;              // [N E K]
  double* Nid = etau + N * E * K;          // [K K] N+ (constant for these models)
  double* rP = Nid + bsz;                  // [N K K] dtau_k/dq_{k+1}, column-major blocks
  double* rT = rP + N * bsz;               // dtau_k/dq_k
  double* rM = rT + N * bsz;               // dtau_k/dq_{k-1} (zero for k < 2, as the assembly wants it)
  double* wR = rM + N * bsz;               // diagonals: R', Qv', Qfv', Qq', Qfq' (TO.cc:1103-1107, pre-scaled on the host)
  double* wQV = wR + K;
  double* wQFV = wQV + K;
  double* wQ = wQFV + K;
  double* wFQ = wQ + K;
  double* mblob = wFQ + K + (K & 1);       // [M.fast_n] the fast shape's records
  double* hA = mblob + M.fast_n + (M.fast_n & 1);   // [(N + 1) K K] x 3: the bands, where the solver stages them from
  double* hB = hA + (N + 1) * qq;
  double* hC = hB + (N + 1) * qq;
  double* hg = hC + (N + 1) * qq;                   // [(N + 1) K]
  double* qn = hg + (N + 1) * K;                    // [(N + 1) K] q_nom, v_nom
  double* vn = qn + (N + 1) * K;
  double* wQf = vn + (N + 1) * K;                   // [K K] x 2: Qq', Qfq' (the C blocks start from their entries)
  double* wFQf = wQf + qq;
  double* cw0 = wFQf + qq + (qq & 1);               // [5 K] the cost's diagonals: Qq, Qv, R, Qfq, Qfv as the YAML has them
  double* cterms = cw0 + 5 * K + (K & 1);           // [3 N + 2] the cost's terms
  double* ccols = cterms + 3 * N + 2;               // [(3 N + 2) K] ... their columns
  double* cp2 = ccols + (3 * N + 2) * K;            // [2 (N + 1) + 2] dq.dq, g~.D^-1 dq per block row (tr_iter_kernel), their sums
  double* cst = cp2 + 2 * (N + 1) + 2;              // [24] the nine inner products and the loop's state words, as the launch found them
  double* kA = cst + 24;                            // (NU > 0) [(N + 1) KK KK] x 3: the KKT bands, [(N + 1) KK] its right-hand side
  double* kB = kA + (NU > 0 ? (N + 1) * KK * KK : 0);
  double* kC = kB + (NU > 0 ? (N + 1) * KK * KK : 0);
  double* kr = kC + (NU > 0 ? (N + 1) * KK * KK : 0);
  double* fz = lds;   // [gn_small_fold_doubles <= S.lds_small] the folded iteration's arrays: in the band solver's carve-up, which
                      // is padded and filled long after them
  int* colinfo = reinterpret_cast<int*>(kr + (NU > 0 ? (N + 1) * KK : 0));   // [K] non-zero rows of N+ column c, [K] non-zero columns of row r
  int* rowinfo = colinfo + K;

  auto stamp = [&](int i) { if (S.ts && tid == 0 && blockIdx.y == 0) S.ts[i] = (double)wall_clock64(); };
  stamp(0);
  TrDecideArgs T = S.T;
  if (!TR) T.state = nullptr;
  const int tau_only = TR ? S.tau_only : 0;
  double dS[11], dst[TRS_COUNT], p2v = 0.0;
  if (TR && T.state) {   // (what the decision needs and nothing in this launch writes: requested first - cost_kernel does the same -,
                   // parked in LDS behind phase A's loads: they would not survive the evaluation in registers)
    T.state = at_problem(T.state, o); T.out = at_problem(T.out, o); T.q = at_problem(T.q, o);
    T.rows += (size_t)blockIdx.y * T.rows_stride; T.part2 = at_problem(T.part2, o);
    if (T.lambda) T.lambda = at_problem(T.lambda, o);
    dS[9] = dS[10] = 0.0;
    if (!S.I.state) {
#pragma unroll
      for (int k = 0; k < 9; ++k) dS[k] = (tid == 0) ? T.out[k] : 0.0;
#pragma unroll
      for (int k = 0; k < TRS_COUNT; ++k) dst[k] = (tid == 0) ? T.state[k] : 0.0;
      if (tid < 2 * T.nblk) p2v = T.part2[tid];
    }
  }
  const bool fold = TR && S.I.state != nullptr;
  bool fold_idle = false;
  if constexpr (TR) if (fold) {
    // ---- T: tr_iter_kernel's part for the iterate (trust_region.h: tr_prepare_rows_body's expressions and orders, the
    // block rows' sums in order, tr_conv_dogleg, the trial point) - every block row in this one workgroup
    TrIterArgs I = S.I;
    {
      TrRowsArgs& A = I.rows;
      A.HA = HA; A.HB = HB; A.HC = HC; A.g = g;
      if (A.jtl) A.jtl = at_problem(A.jtl, o);
      A.yin = at_problem(A.yin, o); A.q = at_problem(A.q, o); A.Dprev = at_problem(A.Dprev, o); A.D = at_problem(A.D, o);
      A.gt = at_problem(A.gt, o); A.w = at_problem(A.w, o); A.part_ll = at_problem(A.part_ll, o);
      if (A.lambda) A.lambda = at_problem(A.lambda, o);
      if (A.freeze) A.freeze = at_problem(A.freeze, o);
      A.slab = at_problem(A.slab, o + (size_t)alt_offset(I.alt, o));   // (the ITERATE's set)
      I.out = at_problem(I.out, o); I.state = at_problem(I.state, o);
      I.q_trial = at_problem(I.q_trial, o); I.dq = at_problem(I.dq, o);
      I.conv.rows += (size_t)blockIdx.y * I.rows_stride;
      if (A.kx.z) {
        A.kx.z = at_problem(A.kx.z, (size_t)blockIdx.y * I.kstride);
        A.kx.w_out = at_problem(A.kx.w_out, o); A.kx.jtl_out = at_problem(A.kx.jtl_out, o); A.kx.lambda_out = at_problem(A.kx.lambda_out, o);
      }
    }
    const TrRowsArgs& A = I.rows;
    const int nvar = (N + 1) * K, nblk = N + 1, nu = A.nu, method = A.scaling_method;
    double* fD = fz;                  // [nvar] each: D, g~, w, D g~, y, g + J^T lambda, q of the iterate, dq_old
    double* fgt = fD + nvar;
    double* fw = fgt + nvar;
    double* fxt = fw + nvar;
    double* fy = fxt + nvar;
    double* fgm = fy + nvar;
    double* fq = fgm + nvar;
    double* fdqo = fq + nvar;
    double* fpt = fdqo + nvar;        // [5 nvar] x 2: the band blocks' partial products, (block row, band block, row)
    double* fpy = fpt + 5 * nvar;
    double* fsv = fpy + 5 * nvar;     // [nblk][8][K]: lanes r < K of a block row's sums 0 .. 6 and 9
    double* fh = fsv + 8 * nvar;      // [nblk][4] x 2: h h and h lambda of the block row's constrained degrees of freedom (lanes)
    double* fhl = fh + 4 * nblk;
    double* fpart = fhl + 4 * nblk;   // [nblk][TR_NSUM]
    double* fS = fpart + 10 * nblk;   // [TR_NSUM], then [a, b, flags], the reader's three words
    double* fab = fS + TR_NSUM;
    double* fwd = fab + 3;
    double st[TRS_COUNT];
#pragma unroll
    for (int k = 0; k < TRS_COUNT; ++k) st[k] = (tid == 0) ? I.state[k] : 0.0;
    const bool frozen = A.freeze && *A.freeze != 0.0;
    for (int e = tid; e < 8 * nblk; e += nt) fh[e] = 0.0;   // (fh, fhl)
    for (int v = tid; v < nvar; v += nt) {
      const int t = v / K, r = v - t * K;
      const double d = (method >= 0) ? scale_factor(method, HC[(size_t)t * qq + r * K + r], A.Dprev[v]) : 1.0;
      double jt = 0.0, yv;
      if (A.kx.z) {   // (kkt_extract_kernel's sums: ascending time step, then dof)
        for (int sx = (t >= 1 ? t - 1 : 0); sx <= t + 1 && sx < N; ++sx)
          for (int jj = 0; jj < nu; ++jj)
            jt += tr_jac_entry(A.slab, A.slab_stride, K, A.kx.nv, sx, A.dofs[jj], N, v) * A.kx.z[(size_t)(sx + 1) * A.kx.KK + K + jj];
        yv = -A.kx.z[(size_t)t * A.kx.KK + r];
        A.kx.w_out[v] = yv; A.kx.jtl_out[v] = jt;
      } else {
        if (A.jtl) jt = A.jtl[v];
        yv = A.yin[v];
      }
      const double gm = (A.jtl || A.kx.z) ? g[v] + jt : g[v];
      const double gti = d * gm, yi = A.ysign * yv;
      fD[v] = d; fgt[v] = gti; fw[v] = yi / d; fxt[v] = d * gti; fy[v] = yi; fgm[v] = gm;
      fq[v] = A.q[v];
      fdqo[v] = I.conv.on ? I.dq[v] : 0.0;
    }
    if (A.kx.z)
      for (int e = tid; e < N * nu; e += nt) {   // lambda_{t-1}: the multiplier rows of z_t
        const int t = 1 + e / nu, j = e - (t - 1) * nu;
        A.kx.lambda_out[e] = A.kx.z[(size_t)t * A.kx.KK + K + j];
      }
    __syncthreads();
    // band block j multiplies x_{i-2+j}: A_i, B_i, C_i, B_{i+1}^T, A_{i+2}^T (blocks column-major)
    for (int e = tid; e < 5 * nvar; e += nt) {
      const int i = e / (5 * K), rem = e - i * 5 * K, j = rem / K, r = rem - j * K, bi = i - 2 + j;
      double at = 0.0, ay = 0.0;
      if (bi >= 0 && bi < nblk) {
        const double* Mb = (j == 0) ? HA + (size_t)i * qq : (j == 1) ? HB + (size_t)i * qq : (j == 2) ? HC + (size_t)i * qq
                         : (j == 3) ? HB + (size_t)(i + 1) * qq : HA + (size_t)(i + 2) * qq;
        const int sr = (j <= 2) ? 1 : K, sc = (j <= 2) ? K : 1;   // M(r, c) or M(c, r)
        double m[K];
#pragma unroll
        for (int c = 0; c < K; ++c) m[c] = Mb[r * sr + c * sc];
#pragma unroll
        for (int c = 0; c < K; ++c) { at += m[c] * fxt[bi * K + c]; ay += m[c] * fy[bi * K + c]; }
      }
      fpt[e] = at; fpy[e] = ay;
    }
    __syncthreads();
    for (int v = tid; v < nvar; v += nt) {   // lane r of block row i: s[0..6], s[9]
      const int i = v / K, r = v - i * K;
      const double* pt = fpt + i * 5 * K;
      const double* py = fpy + i * 5 * K;
      const double d = fD[v];
      const double Hg = d * ((((pt[r] + pt[K + r]) + pt[2 * K + r]) + pt[3 * K + r]) + pt[4 * K + r]);
      const double Hw = d * ((((py[r] + py[K + r]) + py[2 * K + r]) + py[3 * K + r]) + py[4 * K + r]);
      const double gti = fgt[v], wi = fw[v], qi = fq[v];
      double* sv = fsv + i * 8 * K + r;
      sv[0] = gti * gti; sv[K] = gti * Hg; sv[2 * K] = wi * wi; sv[3 * K] = gti * wi; sv[4 * K] = gti * Hw; sv[5 * K] = wi * Hw;
      sv[6 * K] = qi * qi; sv[7 * K] = fgm[v] * fdqo[v];
    }
    if (nu > 0)
      for (int e = tid; e < N * nu; e += nt) {   // lane j of block row i < N: h = tau_i[unactuated] (TO.cc:1274-1278)
        const int i = e / nu, j = e - i * nu;
        const double h = A.slab[(size_t)i * A.slab_stride + A.tau_off + A.dofs[j]];
        fh[4 * i + j] = h * h;
        if (A.kx.z) fhl[4 * i + j] = h * A.kx.z[(size_t)(i + 1) * A.kx.KK + K + j];
        else if (A.lambda) fhl[4 * i + j] = h * A.lambda[i * nu + j];
      }
    __syncthreads();
    // a block row's sums: tr_prepare_rows_body's tree over the 32 lanes of which at most four hold anything but 0.0 -
    // x_l + 0.0, then (x_0 + x_2) + (x_1 + x_3), then 0.0 + that (the levels above add 0.0 to a sum that is not -0.0)
    for (int e = tid; e < 10 * nblk; e += nt) {
      const int i = e / 10, k = e - i * 10;
      const double* src = (k == 7) ? fh + 4 * i : (k == 8) ? fhl + 4 * i : fsv + (i * 8 + (k == 9 ? 7 : k)) * K;
      const int lanes = (k == 7 || k == 8) ? 4 : K;
      double x[4];
#pragma unroll
      for (int l = 0; l < 4; ++l) x[l] = (l < lanes ? src[l < lanes ? l : 0] : 0.0) + 0.0;
      fpart[e] = 0.0 + ((x[0] + x[2]) + (x[1] + x[3]));
    }
    if (tid < 3) {   // the reader's words (it started with this workgroup)
      double x = 0.0;
      unsigned polls = 0;
      while (!tr_ll_try(A.part_ll + 2 * (TR_NSUM * nblk + tid), A.epoch, x)) {
        __builtin_amdgcn_s_sleep(1);
        if (++polls > (1u << 17)) { x = (tid == 1) ? (double)I.fact_id : 0.0; break; }   // (reported as a timeout)
      }
      fwd[tid] = x;
    }
    __syncthreads();
    if (tid < TR_NSUM) {
      double acc = 0.0;
      for (int b = 0; b < nblk; ++b) acc += fpart[b * TR_NSUM + tid];
      fS[tid] = acc;
      if (tid < 9) I.out[tid] = acc;
    }
    __syncthreads();
    const bool singular = fwd[2] != 0.0;
    if (tid == 0) {
      int extra = 0;
      if (I.fact_status && (unsigned)fwd[0] == I.fact_id) extra |= TRF_FACTORIZATION;
      if (I.timeout_status && (unsigned)fwd[1] == I.fact_id) extra |= TRF_SOLVER_TIMEOUT;
      if (singular) extra |= TRF_SINGULAR_S;
      tr_conv_dogleg(I, fS, st, true, extra, fab);
#pragma unroll
      for (int k = 0; k < 9; ++k) cst[k] = fS[k];   // (what the decision below takes from tr_iter_kernel's launch)
#pragma unroll
      for (int k = 0; k < TRS_COUNT; ++k) cst[9 + k] = st[k];
    }
    __syncthreads();
    const double a = fab[0], b = fab[1];
    fold_idle = fab[2] != 0.0;
    for (int v = tid; v < nvar; v += nt) {
      const double dqs = a * fgt[v] + b * fw[v];
      const double dq = I.scaling ? fD[v] * dqs : dqs;
      if (!fold_idle) { I.dq[v] = dq; I.q_trial[v] = fq[v] + dq; }
      qs[v] = fold_idle ? I.q_trial[v] : fq[v] + dq;   // (an idling loop keeps the trial point of its last decided iteration)
      fpt[v] = dq * dq; fpy[v] = fgt[v] * dqs;
      if (!frozen && !singular) {
        A.D[v] = fD[v]; A.gt[v] = fgt[v]; A.w[v] = fw[v];
        const_cast<double*>(A.Dprev)[v] = fD[v];
      }
    }
    __syncthreads();
    for (int i = tid; i < nblk; i += nt) {   // the row's dq.dq and g~.(a g~ + b w), in row order
      double x0 = 0.0, x1 = 0.0;
#pragma unroll
      for (int r = 0; r < K; ++r) { x0 += fpt[i * K + r]; x1 += fpy[i * K + r]; }
      cp2[2 * i] = x0; cp2[2 * i + 1] = x1;
    }
  }
  // ---- A: every global load of the step (but the nominal trajectory, first used in E)
  for (int i = tid; i < M.fast_n; i += nt) mblob[i] = M.blob[M.fast_lo + i];
  if (!fold)
    for (int i = tid; i < (N + 1) * K; i += nt) qs[i] = q[i];
  for (int i = tid; i < bsz; i += nt) { Nid[i] = M.nplus_const[i]; wQf[i] = P.Qq[i]; wFQf[i] = P.Qfq[i]; }
  for (int i = tid; i < (N + 1) * K; i += nt) { qn[i] = P.q_nom[i]; vn[i] = P.v_nom[i]; }
  const BandLds BL = band_layout(S.B.n * KK, WS);
  band_pad(lds, BL, tid, nt);   // (the solver's copies: their padding now, behind this phase's barrier)
  if (tid < K) {
    wR[tid] = P.R[tid * nv + tid]; wQV[tid] = P.Qv[tid * nv + tid]; wQFV[tid] = P.Qfv[tid * nv + tid];
    wQ[tid] = P.Qq[tid * nq + tid]; wFQ[tid] = P.Qfq[tid * nq + tid];
    colinfo[tid] = M.colinfo[tid];
    rowinfo[tid] = M.rowinfo[tid];
    if (TR && T.state) {
      cw0[tid] = P.Qq0[tid * nq + tid]; cw0[K + tid] = P.Qv0[tid * nv + tid]; cw0[2 * K + tid] = P.R0[tid * nv + tid];
      cw0[3 * K + tid] = P.Qfq0[tid * nq + tid]; cw0[4 * K + tid] = P.Qfv0[tid * nv + tid];
    }
  }
  if (TR && T.state && !fold) {
    if (tid == 0) {
#pragma unroll
      for (int k = 0; k < 9; ++k) cst[k] = dS[k];
#pragma unroll
      for (int k = 0; k < TRS_COUNT; ++k) cst[9 + k] = dst[k];
    }
    if (tid < 2 * T.nblk) cp2[tid] = p2v;
    for (int idx = tid + nt; idx < 2 * T.nblk; idx += nt) cp2[idx] = T.part2[idx];
  }
  __syncthreads();
  stamp(1);
  const DevModel Ml = rebase_model(M, mblob - M.fast_lo);

  // ---- B: v_t = N+ (q_t - q_{t-1}) / dt (fd_body's sparse form: the row's non-zero range), then a_k and the perturbations
  for (int idx = tid; idx < (N + 1) * K; idx += nt) {
    const int t = idx / K, r = idx - t * K;
    double vr;
    if (t == 0) {
      vr = P.v_init[r];
    } else {
      const int ri = rowinfo[r], c0 = ri & 0xffff, c1 = c0 + (ri >> 16);
      double acc = Nid[c0 * nv + r] * (qs[t * K + c0] - qs[(t - 1) * K + c0]);
      for (int c = c0 + 1; c < c1; ++c) acc += Nid[c * nv + r] * (qs[t * K + c] - qs[(t - 1) * K + c]);
      vr = acc / dt;
    }
    vs[idx] = vr;
    v_out[idx] = vr;
  }
  const double eps = 1.4901161193847656e-08;  // sqrt(DBL_EPSILON)
  for (int idx = tid; idx < N * E; idx += nt) {
    const int k = idx / E, e = idx - k * E;
    double dq = 1.0;
    if (e >= 1 && e < 1 + nP + nT) {
      const int i = (e - 1) % nq;
      const double qi = (e < 1 + nP) ? qs[(k + 1) * K + i] : qs[k * K + i];
      dq = eps * __builtin_fmax(1.0, __builtin_fabs(qi));
      const double temp = qi + dq;
      dq = temp - qi;
    }
    edq[idx] = dq;
    const double dv = dq / dt;
    edv[idx] = dv;
    eda[idx] = dv / dt;
  }
  for (int i = tid; i < (N + 1) * bsz; i += nt) nplus_out[i] = Nid[i % bsz];
  __syncthreads();
  for (int idx = tid; idx < N * K; idx += nt) {
    const double ar = (vs[idx + K] - vs[idx]) / dt;
    as[idx] = ar;
    a_out[idx] = ar;
  }
  __syncthreads();
  stamp(2);

  // ---- C: the evaluations, lane = (k, e)
  {
    FastTab FT;
    FT.body = Ml.f_body; FT.cbody = Ml.f_cbody; FT.pairs = Ml.f_pairs; FT.seg = Ml.f_seg; FT.maxpp = Ml.f_maxpp;
    for (int idx = tid; idx < N * E; idx += nt) {
      const int k = idx / E, el = idx - k * E;
      if (tau_only && el != 0) continue;
      InFwd in;
      in.q1 = qs + (k + 1) * K; in.v1 = vs + (k + 1) * K; in.a0 = as + k * K; in.N1 = Nid; in.N0 = Nid; in.nv = nv;
      in.kind = (el == 0) ? 0 : ((el < 1 + nP) ? 1 : ((el < 1 + nP + nT) ? 2 : 3));
      in.col = (el == 0) ? 0 : ((el < 1 + nP) ? el - 1 : ((el < 1 + nP + nT) ? el - 1 - nP : el - 1 - nP - nT));
      in.dq = edq[idx];
      in.sdv = (in.kind == 2) ? -edv[idx] : ((in.kind == 1) ? edv[idx] : 0.0);
      in.sda = (in.kind == 2) ? -eda[idx] : ((in.kind == 1) ? eda[idx] : 0.0);
      in.keep = (in.kind == 3) ? 0ull : ~0ull;
      in.keep0 = (in.kind == 2) ? ~0ull : 0ull;
      id_eval_fast<FS::MAXC, FS::NP, FS::CJ, FS::J0, FS::K0, FS::W2>(FT, Ml.gravity, S.cp, 0, el < 1 + nP + nT, in, etau + idx * K);
    }
  }
  __syncthreads();
  stamp(3);

  // ---- D: the records (fd_body, mode 1)
  {
    const double sc = 1 / dt / dt;
    for (int idx = tid; idx < (tau_only ? 0 : N * bsz); idx += nt) {
      const int k = idx / bsz, rem = idx - k * bsz, i = rem / nv, r = rem - i * nv;
      const double* et = etau + k * E * K;
      const int ci = colinfo[i], j0 = ci & 0xffff, cnt = ci >> 16;
      const double tp = et[(1 + i) * nv + r], tt = et[(1 + nP + i) * nv + r], t0 = et[r];
      const double dqp = edq[k * E + 1 + i], dqt = edq[k * E + 1 + nP + i];
      const double* Mcols = et + (1 + nP + nT) * nv;
      const double pv = (tp - t0) / dqp;
      const double tv = (k >= 1) ? (tt - t0) / dqt : 0.0;
      double acc = (sc * Mcols[j0 * nv + r]) * Nid[i * nv + j0];
      if (cnt > 1) acc += (sc * Mcols[(j0 + 1) * nv + r]) * Nid[i * nv + j0 + 1];
      if (cnt > 2) acc += (sc * Mcols[(j0 + 2) * nv + r]) * Nid[i * nv + j0 + 2];
      const double fillM = (k == 0) ? __builtin_nan("") : 0.0;
      double* sl = slab + (size_t)k * S.slab_stride;
      sl[rem] = (k >= 2) ? acc : fillM;          // dtau_k/dq_{k-1}
      sl[bsz + rem] = tv;                        // dtau_k/dq_k
      sl[2 * bsz + rem] = pv;                    // dtau_k/dq_{k+1}
      rP[idx] = pv; rT[idx] = tv; rM[idx] = (k >= 2) ? acc : 0.0;
    }
    for (int idx = tid; idx < N * nv; idx += nt) {
      const int k = idx / nv, r = idx - k * nv;
      slab[(size_t)k * S.slab_stride + 3 * bsz + r] = etau[k * E * K + r];
    }
  }
  __syncthreads();
  stamp(4);

  // ---- (trust-region loop) the cost of the trial point and the decision: cost_kernel's items, sums and order
  // (kernels.h; diagonal weights: e^T W e per term as sum_c ((0 + e_c w_c) e_c), the columns in order, the terms in order)
  if constexpr (TR) if (T.state) {
    for (int idx = tid; idx < (N + 1) * K; idx += nt) {
      const int t = idx / K, c = idx - t * K;
      const bool run = t < N;
      const double eq = qs[idx], nq_ = qn[idx], ev = vs[idx], nv_ = vn[idx];
      const double et = run ? etau[t * E * K + c] : 0.0;
      double valq, valv, valt;
      { const double dc = eq - nq_; double acc = 0; acc += dc * (run ? cw0[c] : cw0[3 * K + c]); valq = acc * dc; }
      { const double dc = ev - nv_; double acc = 0; acc += dc * (run ? cw0[K + c] : cw0[4 * K + c]); valv = acc * dc; }
      { const double dc = et - 0.0; double acc = 0; acc += dc * cw0[2 * K + c]; valt = acc * dc; }
      const int term = run ? 3 * t : 3 * N;
      ccols[term * K + c] = valq;
      ccols[(term + 1) * K + c] = valv;
      if (run) ccols[(term + 2) * K + c] = valt;
    }
    __syncthreads();
    for (int term = tid; term < 3 * N + 2; term += nt) {
      double tot = 0;
#pragma unroll
      for (int c = 0; c < K; ++c) tot += ccols[term * K + c];
      cterms[term] = tot;
    }
    __syncthreads();
    if (tid == 64 || tid == 128) {   // (beside thread 0's chain of adds)
      const int k = tid == 64 ? 0 : 1;
      double acc = 0.0;
      for (int b = 0; b < T.nblk; ++b) acc += cp2[2 * b + k];
      cp2[2 * T.nblk + k] = acc;
    }
    double cost = 0;
    if (tid == 0) {
      for (int i = 0; i < 3 * N; ++i) cost += cterms[i];
      cost *= P.dt;
      cost += cterms[3 * N];
      cost += cterms[3 * N + 1];
      *at_problem(S.cost_out, o) = cost;
    }
    const int neq = T.nu * T.N;
    __syncthreads();
    if (T.nu > 0) {   // h(q + dq) . lambda: products by everybody, added in index order by thread 0
      for (int r = tid; r < neq; r += nt) {
        const int t = r / T.nu, j = r - t * T.nu;
        ccols[r] = etau[t * E * K + T.dofs[j]] * T.lambda[r];
      }
      __syncthreads();
    }
    if (tid == 0) {
#pragma unroll
      for (int k = 0; k < 9; ++k) dS[k] = cst[k];
#pragma unroll
      for (int k = 0; k < TRS_COUNT; ++k) dst[k] = cst[9 + k];
      dS[9] = cp2[2 * T.nblk]; dS[10] = cp2[2 * T.nblk + 1]; T.out[9] = dS[9]; T.out[10] = dS[10];
      double hl = 0.0;
      for (int r = 0; r < (T.nu > 0 ? neq : 0); ++r) hl += ccols[r];
      cterms[0] = tr_decide(T, cost, hl, dS, dst) ? 1.0 : 0.0;
    }
    __syncthreads();
    const bool accepted = cterms[0] != 0.0;
    if (accepted)
      for (int idx = tid; idx < (N + 1) * K; idx += nt) T.q[idx] = qs[idx];
    if (!accepted || tau_only) return;   // (a rejected step keeps g, H and the step of the iterate)
  }

  // ---- E: g and the bands (assemble_diag_body: same terms, same order).  A thread per (block row, r, c) forms C_i(r, c),
  // B_i(r, c), A_i(r, c) from the same nine operand columns; a thread per (block row, j) the gradient entry.  Straight-line:
  // every operand is loaded whatever the row (block indices clamped), every sum is formed, and the row's conditions
  // (TO.cc:1127-1161: which terms exist at i = N - 1, N, below 2 / 3) SELECT among them - a branch per condition put an LDS
  // round trip behind every term (4.5 us for acrobot's 574 entries, 11.5 us for the spinner's 1230).  The weighted operand
  // X(l, r) = A(l, r) w_l is formed here, the very product the staged copy of assemble_diag_body holds.
  // (the solver's staging table: this thread's first items, on their way while the sums below are formed)
  constexpr int NST = 6;
  BandStageItem st[NST];
#pragma unroll
  for (int u = 0; u < NST; ++u) {
    const int e = tid + u * nt;
    const int4 raw = *reinterpret_cast<const int4*>(S.stage + (e < S.nstage ? e : 0));
    st[u].src = raw.x; st[u].rhs = raw.y; st[u].dst = e < S.nstage ? raw.z : -1; st[u].dst0 = e < S.nstage ? raw.w : -1;
  }
  {
    const double idt = 1 / dt, midt = -1 / dt;
    const int per = qq + nq;   // items per block row
    for (int idx = tid; idx < (N + 1) * per; idx += nt) {
      const int i = idx / per, e = idx - i * per;
      double* Cg = HC + (size_t)i * qq;
      double* Bg = HB + (size_t)i * qq;
      double* Ag = HA + (size_t)i * qq;
      double* Cl = hC + i * qq;   // (the same entries once more in LDS: the solver's staging reads them there)
      double* Bl = hB + i * qq;
      double* Al = hA + i * qq;
      const int im1 = i >= 1 ? i - 1 : 0, i0 = i < N ? i : N - 1, ip1 = i < N - 1 ? i + 1 : N - 1;
      const double* Pm1 = rP + im1 * bsz;   // S_PM1
      const double* Tm1 = rT + im1 * bsz;   // S_TM1
      const double* Mm1 = rM + im1 * bsz;   // S_MM1 (i >= 3)
      const double* T0 = rT + i0 * bsz;     // S_T0  (i < N)
      const double* M0 = rM + i0 * bsz;     // S_M0  (i < N; rM is zero below k = 2)
      const double* Mp1 = rM + ip1 * bsz;   // S_MP1 (i < N - 1)
      if (e < qq) {
        const int c = e / nq, r = e - c * nq;
        double pr[K], pc[K], tr_[K], tc[K], mr[K], mc[K], tm[K], m0[K], mm[K], nr[K], nc[K], wv[K], ww[K], wr[K];
#pragma unroll
        for (int l = 0; l < K; ++l) {
          pr[l] = Pm1[r * K + l]; pc[l] = Pm1[c * K + l]; tr_[l] = T0[r * K + l]; tc[l] = T0[c * K + l];
          mr[l] = Mp1[r * K + l]; mc[l] = Mp1[c * K + l]; tm[l] = Tm1[c * K + l]; m0[l] = M0[c * K + l]; mm[l] = Mm1[c * K + l];
          nr[l] = Nid[r * K + l]; nc[l] = Nid[c * K + l];
          wv[l] = (i < N) ? wQV[l] : wQFV[l]; ww[l] = (i < N - 1) ? wQV[l] : wQFV[l]; wr[l] = wR[l];
        }
        const double w0 = (i < N) ? wQf[c * nq + r] : wFQf[c * nq + r];
        double xP[K], xT[K], xM[K], xV[K], xW1[K], sV[K], sW[K];
#pragma unroll
        for (int l = 0; l < K; ++l) {
          xP[l] = pr[l] * wr[l]; xT[l] = tr_[l] * wr[l]; xM[l] = mr[l] * wr[l];
          xV[l] = (idt * nr[l]) * wv[l]; xW1[l] = (midt * nr[l]) * ww[l];
          sV[l] = idt * nc[l]; sW[l] = midt * nc[l];
        }
        auto dot = [&](const double (&x)[K], const double (&y)[K]) __attribute__((always_inline)) {
          double acc = x[0] * y[0];
#pragma unroll
          for (int l = 1; l < K; ++l) acc = acc + x[l] * y[l];
          return acc;
        };
        const double dVV = dot(xV, sV), dPP = dot(xP, pc), dTT = dot(xT, tc), dMM = dot(xM, mc), dWW = dot(xW1, sW);
        const double dPT = dot(xP, tm), dTM = dot(xT, m0), dVW = dot(xV, sW), dPM = dot(xP, mm);
        // C_i (TO.cc:1127-1137 / :1157-1161)
        const double c2 = (w0 + dVV) + dPP;
        const double c3 = c2 + dTT;
        const double c4 = (i < N - 1) ? c3 + dMM : c3;
        const double cN = (i < N) ? c4 + dWW : c2;
        // B_i (TO.cc:1140-1147), A_i (:1150-1153)
        const double b2 = (i < N) ? dPT + dTM : dPT;
        const double bN = (i >= 2) ? b2 + dVW : 0.0;
        const double aN = (i >= 3) ? dPM : 0.0;
        const double Cv = (i == 0) ? ((r == c) ? 1.0 : 0.0) : cN;
        const double Bv = (i == 0) ? 0.0 : bN, Av = (i == 0) ? 0.0 : aN;
        if (r >= c) {   // the lower triangle's thread writes both mirror images (MakeSymmetric)
          Cg[c * nq + r] = Cv; Cg[r * nq + c] = Cv;
          Cl[c * nq + r] = Cv; Cl[r * nq + c] = Cv;
        }
        Bg[e] = Bv; Bl[e] = Bv;
        Ag[e] = Av; Al[e] = Av;
      } else {   // gradient block (TO.cc:1046-1080)
        const int j = e - qq;
        double pj[K], tj[K], mj[K], nj[K], ev[K], evp[K], em1[K], e0[K], ep1[K];
#pragma unroll
        for (int r = 0; r < K; ++r) {
          pj[r] = Pm1[j * K + r]; tj[r] = T0[j * K + r]; mj[r] = Mp1[j * K + r]; nj[r] = Nid[j * K + r];
          ev[r] = vs[i * K + r] - vn[i * nv + r];
          evp[r] = vs[(i < N ? i + 1 : N) * K + r] - vn[(i < N ? i + 1 : N) * nv + r];
          em1[r] = etau[im1 * E * K + r]; e0[r] = etau[i0 * E * K + r]; ep1[r] = etau[ip1 * E * K + r];
        }
        const double qe = qs[i * K + j] - qn[i * nq + j];
        // sum_r (e_r w_r) J[r]
        auto vwm = [&](const double (&ee)[K], const double* wgt, const double (&J)[K], const double sj) __attribute__((always_inline)) {
          double acc = (ee[0] * wgt[0]) * (sj * J[0]);
#pragma unroll
          for (int r = 1; r < K; ++r) acc += (ee[r] * wgt[r]) * (sj * J[r]);
          return acc;
        };
        auto vwr = [&](const double (&ee)[K], const double* wgt, const double (&J)[K]) __attribute__((always_inline)) {
          double acc = (ee[0] * wgt[0]) * J[0];
#pragma unroll
          for (int r = 1; r < K; ++r) acc += (ee[r] * wgt[r]) * J[r];
          return acc;
        };
        const double gV = vwm(ev, wQV, nj, idt), gVf = vwm(ev, wQFV, nj, idt);
        const double gW = vwm(evp, (i == N - 1) ? wQFV : wQV, nj, midt);
        const double gP = vwr(em1, wR, pj), gT = vwr(e0, wR, tj), gM = vwr(ep1, wR, mj);
        double ga = qe * wQ[j];
        ga = ga + gV;
        ga = ga + gW;
        ga = ga + gP;
        ga = ga + gT;
        ga = (i != N - 1) ? ga + gM : ga;
        double gb = gP;
        gb = gb + qe * wFQ[j];
        gb = gb + gVf;
        const double gj = (i == 0) ? 0.0 : ((i < N) ? ga : gb);
        g[(size_t)i * nq + j] = gj;
        hg[i * nq + j] = gj;
      }
    }
  }
  // ---- F: the band solve, from the bands just written (this workgroup's own stores: a barrier's release / acquire at
  // workgroup scope orders them; nothing of g or H was read by this compute unit before)
  __syncthreads();
  stamp(5);
  PipeAsm F{};
  BandArgs B = S.B;   // the solver stages from the LDS copies (generic pointers into LDS: no problem offset), writes to memory
  B.pstride = 0;
  B.ts = (S.ts && blockIdx.y == 0) ? S.ts + 8 : nullptr;   // (the solver's own stamps behind this kernel's)
  const double* sbase = hA;   // what the staging table's sources are relative to
  if constexpr (NU > 0) {
    // the KKT system of kkt.h (kkt_build_kernel's entries: M_{t,t-2}, M_{t,t-1}, M_{t,t} with the rows of J out of the
    // records of step t - 1, right-hand side [g_t ; h_{t-1}]) from what this workgroup holds
    constexpr int kk2 = KK * KK;
    for (int idx = tid; idx < (N + 1) * 3 * kk2; idx += nt) {
      const int t = idx / (3 * kk2), rem = idx - t * 3 * kk2, band = rem / kk2, e = rem - band * kk2, c = e / KK, r = e - c * KK;
      const int sc = t - 2 + band;   // block column
      double v = 0.0;
      if (sc >= 0) {
        if (r < K && c < K) {
          v = (band == 0 ? hA : band == 1 ? hB : hC)[t * qq + c * K + r];
        } else if (r >= K && c < K) {   // mu_t's row: J_{t-1, sc}[dof, c]
          const bool zero = t < 1 || (band == 0 && t - 1 < 2) || (band == 1 && t - 1 < 1);
          if (!zero) v = (band == 0 ? rM : band == 1 ? rT : rP)[(t - 1) * bsz + c * nv + T.dofs[r - K]];
        } else if (r < K) {             // mu_sc's column: J_{sc-1, t}^T, the diagonal block's only
          if (band == 2 && t >= 1) v = rP[(t - 1) * bsz + r * nv + T.dofs[c - K]];
        } else if (t == 0 && band == 2 && r == c) {
          v = 1.0;                      // the dummy mu_0
        }
      }
      (band == 0 ? kA : band == 1 ? kB : kC)[t * kk2 + e] = v;
    }
    for (int idx = tid; idx < (N + 1) * KK; idx += nt) {
      const int t = idx / KK, r = idx - t * KK;
      kr[idx] = (r < K) ? hg[t * K + r] : (t >= 1 ? etau[(t - 1) * E * K + T.dofs[r - K]] : 0.0);   // h = tau_{t-1}[dof]
    }
    __syncthreads();
    const size_t ok = (size_t)blockIdx.y * S.kstride;
    B.x = at_problem(B.x, ok); B.Dst = at_problem(B.Dst, ok);
    B.HA = kA + S.kkt_r0 * kk2; B.HB = kB + S.kkt_r0 * kk2; B.HC = kC + S.kkt_r0 * kk2; B.b = kr + S.kkt_r0 * KK;
    sbase = kA;
  } else {
    B.x = at_problem(B.x, o); B.Dst = at_problem(B.Dst, o);
    B.HA = hA + qq; B.HB = hB + qq; B.HC = hC + qq; B.b = hg + K;   // (from block row 1 on: row 0 is the identity)
  }
  {   // the solver's two copies of the band, by the table (penta_band_body's staging loop, its index arithmetic done once on the host)
    auto put = [&](const BandStageItem& it) __attribute__((always_inline)) {
      const double raw = sbase[it.src >= 0 ? it.src : 0];
      const double val = it.src >= 0 ? raw * (it.rhs ? B.rhs_sign : 1.0) : 0.0;
      if (it.dst >= 0) lds[it.dst] = val;
      if (it.dst0 >= 0) lds[it.dst0] = val;
    };
#pragma unroll
    for (int u = 0; u < NST; ++u) put(st[u]);
    for (int e = tid + NST * nt; e < S.nstage; e += nt) put(S.stage[e]);
  }
  __syncthreads();
  penta_band_body<WS, true, true>(B, F);
  stamp(6);
}

}  // namespace idto_dev
