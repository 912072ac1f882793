// penta_ldl.h — block LDL^T factor + solve of the symmetric positive-definite block
// penta-diagonal Gauss-Newton Hessian: the production replacement of the reference's
// PentaDiagonalFactorization (optimizer/penta_diagonal_solver.h:124-248).
//
// Same block recursion as the reference's Thomas algorithm [Benkert & Fischer 2007],
// written for the symmetric case (D_i = B_{i+1}^T, E_i = A_{i+2}^T):
//   S_i  = C_i - Et_{i-2}^T Dn_{i-2} Et_{i-2} - Ht_{i-1}^T Dn_{i-1} Ht_{i-1}      (reference: G_i)
//   H_i  = B_{i+1}^T - Ht_{i-1}^T Dn_{i-1} Et_{i-1}                               (reference: D_i - K_i Z_{i-1})
//   y_i  = r_i - Ht_{i-1}^T Dn_{i-1} rt_{i-1} - Et_{i-2}^T Dn_{i-2} rt_{i-2}
//   S_i = L_i D_i L_i^T (scalar elimination, no pivoting):  U_i = D_i L_i^T,  Dn_i = D_i^{-1},
//   [Ht_i | Et_i | rt_i] = L_i^{-1} [H_i | A_{i+2}^T | y_i]
//   backward:  U_i x_i = rt_i - Ht_i x_{i+1} - Et_i x_{i+2}
// i.e. the reference's Y_i = S_i^{-1} H_i is kept in the factored form L_i^{-T} Dn_i Ht_i:
// this halves the dependent elimination steps per block row and — measured on the hopper,
// cond(H) ~ 1e10 — is what keeps the residual at the level of the reference's pivoted LU
// (forming S^{-1} H explicitly without pivoting lost 4 digits).  Differences from the
// reference are at round-off level: no pivoting inside S_i (it is SPD), FMAs, reciprocals by
// v_rcp_f64 + 2 Newton steps.  `penta_kernel` (kernels.h) stays the bit-exact restatement.
//
// Hardware mapping (workgroups of 4 wavefronts; the recursion over i is sequential, the kernel is
// bound by the instruction issue rate of the wavefront on the critical path, ~5 cycles per VALU
// instruction and 7 / 34 cycles per LDS read / write, so the design minimises instructions there):
//   * two workgroups eliminate from both ends of the horizon and meet at block rows m, m+1
//     ("twisted" factorisation, see the kernel); the dependent chain is n/2 block rows;
//   * elimination: the augmented block [S_i | H_i | E_i | y] lives in the REGISTERS of one
//     wavefront per (64 - K) right-hand-side columns, one column per lane; a pivot step
//     broadcasts the pivot column with v_readlane (wave-uniform values sit in SGPRs) and is
//     otherwise per-lane FMAs: no LDS traffic and no barrier in the K dependent steps;
//   * block products X^T Dn Y on the matrix cores (v_mfma_f64_16x16x4): every operand element is
//     read from LDS once per 16x16 tile; while wavefront 0 eliminates, the others form
//     Et^T Dn Et for the next row (MFMA), stage and prefetch the next rows' bands and write
//     factors back to HBM;
//   * barriers order LDS only (s_waitcnt lgkmcnt(0); s_barrier): the register prefetch of the
//     next row's blocks and the write-backs stay in flight across them;
//   * back substitution in "push" form: every component of x_i is pushed into the pending
//     right-hand sides of rows i-1, i-2 as soon as the triangular solve produces it.
// Many right-hand sides: penta_apply.h (one wavefront per column, from the stored factors).
#pragma once

#include <hip/hip_runtime.h>

#include "batch.h"

namespace idto_dev {

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ double rdlane(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// v[LANE] = val (val wave-uniform, LANE an inline constant: one SGPR operand at most); clang has
// no builtin for v_writelane_b32
template <int LANE>
__device__ __forceinline__ void wrlane(int& v, int val) {
  asm("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(val), "n"(LANE));
}

// 1/d to full double precision: hardware estimate + 2 Newton steps (5 instructions instead of
// the ~15 of an IEEE division; the factorisation is not bit-compared with the host).
__device__ __forceinline__ double fast_rcp(double d) {
  double x = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, x, 1.0);
  x = __builtin_fma(x, e, x);
  e = __builtin_fma(-d, x, 1.0);
  x = __builtin_fma(x, e, x);
  return x;
}

// ---- bounded waits between workgroups.  The multi-workgroup solvers (two-sided, nested dissection, the
// pipelined chains, the fused Gauss-Newton launch) synchronise through flags in global memory and need their
// partners resident at the same time - true by head-count on an otherwise idle device, not by construction
// when other contexts' kernels occupy compute units.  Every such wait gives up after SPIN_LIMIT_TICKS of the
// 100 MHz wall clock and reports it in host-mapped memory ([0] = id of the launch, [1] = count); the kernel
// then runs to its end on whatever it has, and the host (FactorStatus in idto_hip.hip) repeats the solve on a
// variant with fewer workgroups.  Nothing can hang the device.
constexpr long long SPIN_LIMIT_TICKS = 5000000;   // 50 ms
struct SpinCtl { unsigned* word; unsigned id; };
template <class Ready>
__device__ __forceinline__ bool spin_wait(Ready ready, const SpinCtl sc) {
  unsigned n = 0;
  long long t0 = 0;
  while (!ready()) {
    __builtin_amdgcn_s_sleep(1);
    if (((++n) & 1023u) == 0) {
      if (sc.word && __hip_atomic_load(sc.word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == sc.id) return false;   // somebody gave up already
      const long long now = (long long)wall_clock64();
      if (t0 == 0) t0 = now;
      else if (now - t0 > SPIN_LIMIT_TICKS) {
        if (sc.word) {
          __hip_atomic_store(sc.word, sc.id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_fetch_add(sc.word + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return false;
      }
    }
  }
  return true;
}

// One pivot step (J is a template parameter so that v_writelane can take the lane as an inline
// constant).  The pivots d_J are wave-uniform (SGPR pairs); lane J keeps its own with v_writelane
// (2 instructions per pivot, no compare/select chain) and inverts it once at the end.
// Software-pipelined: the reciprocal of pivot J+1 (broadcast + v_rcp + 2 Newton steps, ~60 cycles
// of dependent latency) only needs row J+1 of the update by pivot J, so that row is updated first
// and the reciprocal chain overlaps with the remaining FMAs of pivot J.
// YP ("y-push", single right-hand side held by lane 3K): the column lanes also accumulate
// sum_j x[j][c] * (rt_j / d_j), i.e. lane K + c ends with (Ht_i^T Dn rt_i)[c] and lane 2K + c with
// (Et_i^T Dn rt_i)[c] - the contributions of this block row to the right-hand sides of the next
// two rows, obtained for 3 instructions per pivot instead of a separate mat-vec phase.
template <int K, int J, bool YP>
__device__ __forceinline__ void ldl_pivot(double (&xr)[K], double d, double inv, int& dlo, int& dhi, double& yacc) {
  wrlane<J>(dlo, __double2loint(d));
  wrlane<J>(dhi, __double2hiint(d));
  const double t = xr[J] * inv;
  if constexpr (YP) {
    const double trhs = rdlane(t, 3 * K);  // (Dn rt)_J: row J of the right-hand side is final here
    yacc = __builtin_fma(xr[J], trhs, yacc);
  }
  // multipliers are wave-uniform (SGPR pairs): broadcast them in chunks so that at most CH pairs
  // are live at a time (the kernel is short of SGPRs; spills cost v_readlane's on this path)
  constexpr int CH = 8;
  double inv_next = 1.0, d_next = 1.0;
  // (multipliers from the UPPER triangle - element (J, r) of the pivot row, not (r, J) of the column: L = (D^-1 U)^T then
  // holds exactly, which is what the back substitution and the spike rows assume; penta_pipe.h does the same)
  if constexpr (J + 1 < K) {
#ifdef IDTO_LDL_LOWER_MULTIPLIERS
    const double m1 = rdlane(xr[J + 1], J);
#else
    const double m1 = rdlane(xr[J], J + 1);
#endif
    xr[J + 1] = __builtin_fma(-m1, t, xr[J + 1]);
    d_next = rdlane(xr[J + 1], J + 1);
    inv_next = fast_rcp(d_next);
  }
#pragma unroll
  for (int r0 = J + 2; r0 < K; r0 += CH) {
    double m[CH];
#pragma unroll
    for (int q = 0; q < CH; ++q)
#ifdef IDTO_LDL_LOWER_MULTIPLIERS
      if (r0 + q < K) m[q] = rdlane(xr[r0 + q], J);
#else
      if (r0 + q < K) m[q] = rdlane(xr[J], r0 + q);
#endif
#pragma unroll
    for (int q = 0; q < CH; ++q)
      if (r0 + q < K) xr[r0 + q] = __builtin_fma(-m[q], t, xr[r0 + q]);
  }
  if constexpr (J + 1 < K) ldl_pivot<K, J + 1, YP>(xr, d_next, inv_next, dlo, dhi, yacc);
}

// Forward elimination of the columns held by this wavefront (lanes < K: columns of S).
// On exit: S lanes hold U = D L^T (upper triangle), rhs lanes L^{-1} rhs; returns 1 / U[l][l]
// in lane l (l < K).
template <int K, bool YP>
__device__ __forceinline__ double ldl_eliminate_wave(double (&xr)[K], int lane, double& yacc) {
  int dlo = 0, dhi = 0x3ff00000;  // 1.0 in the lanes that hold no pivot
  const double d = rdlane(xr[0], 0);
  yacc = 0.0;
  ldl_pivot<K, 0, YP>(xr, d, fast_rcp(d), dlo, dhi, yacc);
  return fast_rcp(__hiloint2double(dhi, dlo));
}

// Column stride (in doubles) of every K-row block kept in LDS and of the row-major factor blocks:
// even (16-byte aligned columns for ds_read_b128), = 2 mod 4, i.e. an odd number of 16-byte
// units, so that consecutive columns start in different LDS bank groups, and >= 4 ceil(K/4): the
// MFMA k-steps read rows up to 4 ceil(K/4) - 1 of a column (zero pad rows).
__host__ __device__ constexpr int ldl_ks(int K) { return 4 * ((K + 3) / 4) + 2; }

struct PentaLdlLds {  // offsets in doubles
  int W, Ht, Et, Iv, rt, U, G, in, dump, yh, ye, Eb, bl, bl_size, xall, end;
  int kks, rts;
};
// `rows` (> 0, single right-hand side): the chain's local rows incl. pseudo-rows - the right-hand side and rt / x of
// every row are indexed by LOCAL row, so a workgroup of the two-sided elimination needs its own half only
__host__ __device__ inline PentaLdlLds penta_ldl_layout(int n, int K, int nrhs, int rows = 0) {
  PentaLdlLds L;
  const int ks = ldl_ks(K), ncr = 2 * K + nrhs;
  L.kks = K * ks;
  L.rts = nrhs * ks;
  int o = 0;
  L.W = o; o += (K + ncr) * ks;   // augmented block [S | H | E | y], column-major, stride ks
  L.Ht = o; o += 2 * L.kks;       // ring: Ht_i, Ht_{i-1}
  L.Et = o; o += 3 * L.kks;       // ring: Et_i, Et_{i-1}, Et_{i-2}
  L.Iv = o; o += 3 * ks;          // ring: 1/diag(U) (padded to ks)
  L.rt = o; o += 3 * L.rts;       // ring: rt_i (forward) / x_i (backward)
  L.U = o; o += 2 * L.kks;        // ring: U_i, U_{i-1} (write-back staging)
  L.G = o; o += (K * K + 1) & ~1; // Et_{i-1}^T Dn Et_{i-1} for the next row (even size: what follows is read as double2;
                                  // b128 reads off a 16-byte boundary halve the LDS throughput)
  {                               // staged A_i, B_{i+1}, C_i, A_{i+2}; reused by the backward pass
    const int fwd = 4 * K * K, bwd = 3 * K * ks + ks;
    L.in = o; o += ((fwd > bwd ? fwd : bwd) + 1) & ~1;
  }
  L.dump = o; o += 2;             // write target of staging lanes without a slot
  L.yh = o; o += 2 * ks;          // y-push rings: (Ht_i^T Dn rt_i) of the last two rows ...
  L.ye = o; o += 3 * ks;          // ... and (Et_i^T Dn rt_i) of the last three
  L.Eb = o; o += 2 * L.kks;       // E_i = A_{i+2}^T staged straight in column layout (row parity)
  L.bl = o;
  const int nr = (rows > 0 && nrhs == 1 && rows < n) ? rows : n;
  L.bl_size = (nrhs * n * K <= 4096) ? nrhs * nr * K : 0;
  o += (L.bl_size + 1) & ~1;      // right-hand sides staged in LDS when small ...
  L.xall = o;                     // ... and rt_i / x_i of every row: [j][n + 2][ks], two leading zero rows
  o += L.bl_size ? nrhs * (nr + 2) * ks : 0;
  L.end = o;
  return L;
}

// Role of one workgroup of the factorisation.  The two-workgroup ("twisted") kernel uses two of
// them: the top one eliminates rows 0 .. m-1 top-down and then the two join rows m, m+1 (joiner),
// the bottom one rows n-1 .. m+2 bottom-up and runs two product-only pseudo-rows whose augmented
// blocks are its contributions to the join rows (producer).  The nested-dissection kernel
// (penta_nd.h) runs four such chains - two producer / joiner pairs around a separator - and needs
// the hooks at the end.
struct ChainCfg {
  int two;        // 0: one workgroup eliminates the whole system
  int mirror;     // 1: local row il is row base - il of the matrix (bands read transposed), 0: base + il
  int producer;   // 1: pseudo-rows + publish; 0: joiner (adds the producer's contributions, eliminates the join rows)
  int base;       // matrix row of local row 0
  int nloc;       // block rows eliminated here (joiner: chain rows + the two join rows)
  int m_split;    // joiner: local index of the first join row
  int dbg_slot;   // which block of the cycle-stamp array
  // nested dissection: a joiner chain publishes, per local row, that its factors (and rt) are in HBM
  // (counter += 1 per I/O wavefront), and subtracts the separator's contribution from rt before its
  // back substitution
  unsigned long long* rowcnt;
  unsigned long long rowcnt_unit;   // what one wavefront adds
  double* rtpub;                    // [local row][K]
  const double* fst; int fstride;   // the spike workgroup's rows [Ft_il | rt_il] (K rows, column stride ks)
  const unsigned long long* frowcnt; unsigned long long frowtarget;   // ... and their per-row release counters
  const double* xsep;               // [x_s | x_{s+1}] once *sepflag == epoch
  unsigned* sepflag;
  const double* xsep_ll;            // the same with the epoch in every word (penta_nd.h ll_store)
  double* ts;                       // optional wall-clock stamps (100 MHz): start, join reached, forward done, backward start, end
  int factor_only;                  // stop once the factors are in HBM: every right-hand side (the first included) goes
                                    // through penta_apply_kernel, the chains' own back substitution is off the path
  SpinCtl spin;                     // where a wait between workgroups that ran out reports it
  int lds_rows;                     // penta_ldl_layout's `rows` (0: every row of the system)
  int npos;                         // > 0: the pivots [npos, k) of every block row belong to multiplier rows of a KKT system
                                    // (kkt.h): negative, and judged by kkt_extract_kernel; 0: a positive definite matrix
  // the seven-workgroup kernel's back substitution in recursion form (penta_pipe.h chain_recursion_tail): the launch has
  // lds_doubles doubles of LDS, a joiner keeps W_il = U_il^-1 Dn Ft_il in wst (rows laid out like fst), the pair's join
  // rows of x change hands through xjoin_ll
  int rec_tail, lds_doubles;
  double* wst;
  double* xjoin_ll;
  // ... and the pair's PRODUCER forms the joiner's W rows (it is idle for longer): the joiner's rows (local row il is
  // matrix row wp_base -/+ il), its spike workgroup's rows and counters, its wst; wrow[il] = epoch once row il of W is there
  int wp_nloc, wp_base, wp_mirror, wp_waves;
  const double* wp_fst; const unsigned long long* wp_frowcnt; double* wp_wst;
  unsigned* wrow;
};
// pivot test of a lane that holds a pivot's 1 / d (`inv`; NaN for d = 0 / inf / NaN) and the diagonal entry the pivot
// started from: positive, finite, and not cancelled to nothing (d <= eps diag0).  Other lanes pass inv = diag0 = 1.
// The multiplier rows of a KKT system are not tested here: a pivot of theirs that vanishes means redundant constraints,
// not an indefinite Hessian, and is reported as such (TRF_SINGULAR_S) from the range of those pivots.
__device__ __forceinline__ bool ldl_pivot_bad(double inv, double diag0, int lane, int k, int npos) {
  if (npos > 0 && lane >= npos && lane < k) return false;
  return !(inv > 0.0 && inv * diag0 < 4503599627370496.0);   // 2^52 = 1 / eps
}
__device__ __forceinline__ void chain_ts(const ChainCfg& cfg, int slot) {
  if (cfg.ts && threadIdx.x == 0) cfg.ts[slot] = (double)wall_clock64();
}

// (defined in penta_pipe.h, next to the pipelined kernel's pipe_backward it runs)
template <int K>
__device__ void chain_recursion_tail(int n, int k, double* x, double* Ust, double* Hst, double* Est, double* Dst,
                                     const ChainCfg& cfg, unsigned epoch, int xall_off);

// ---- what follows the forward elimination of a chain (shared by penta_ldl_body and the pipelined forward pass of
// penta_pipe.h): the nested-dissection correction of rt by the separator's solution, then the back substitution.
// On entry: the chain's factors are in HBM (Ust / Hst / Est row-major with stride ks, rows scaled by 1/d; Dst = 1/d),
// rt_il (raw) of every local row in lds[xall_off + (il + 2) * ks + r] with two leading zero rows, and a block-wide
// barrier has ordered both; `W_off` is LDS scratch of at least 2K doubles.
template <int K, int NT>
__device__ __forceinline__ void
penta_ldl_tail(int n, int k, int nrhs, double* __restrict__ x, const double* __restrict__ Ust,
               const double* __restrict__ Hst, const double* __restrict__ Est, const double* __restrict__ Dst,
               double* __restrict__ dbg, const ChainCfg cfg, double* __restrict__ xch, unsigned* __restrict__ flags,
               unsigned epoch, int xall_off, int bl_size, int W_off, int nfwd) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool two = cfg.two != 0, mirror = cfg.mirror != 0, producer = cfg.producer != 0;
  const int m_split = cfg.m_split, nloc = cfg.nloc;
  auto orig = [&](int il) { const int o = mirror ? cfg.base - il : cfg.base + il; return o < 0 ? 0 : o; };
  constexpr int nt = NT, ks = ldl_ks(K), NW = NT / 64;
  const int ncr = 2 * K + nrhs;
  const size_t nk = (size_t)n * k;
  double* Wm = lds + W_off;
  auto stamp = [&](int i, int ph) {
    if (dbg && lane == 0)
      dbg[((cfg.dbg_slot * (NT / 64) + wave) * (n + 3) + i) * 8 + ph] = (double)__builtin_readcyclecounter();
  };
  if (cfg.fst) {
    // nested dissection: this chain's rows also couple to the separator.  Once it is solved,
    // rt_il -= Ft_il [x_near ; x_far] with the eliminated coupling blocks Ft_il the spike workgroup
    // left in HBM (columns 0..K-1: the separator row next to this chain's first row).  The rows of
    // Ft are fetched while the separator is still being eliminated (this workgroup is idle then).
    const int pidx = (tid < nloc * K) ? tid : 0, pil = pidx / K, pr = pidx - pil * K;
    // (a second element per thread - 16 local rows of 23 are more than the 256 threads - is fetched ahead in the same way)
    const int qidx = (tid + nt < nloc * K) ? tid + nt : 0, qil = qidx / K, qr = qidx - qil * K;
    spin_wait([&] { return __hip_atomic_load(cfg.frowcnt + pil, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= cfg.frowtarget; }, cfg.spin);
    spin_wait([&] { return __hip_atomic_load(cfg.frowcnt + qil, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= cfg.frowtarget; }, cfg.spin);
    (void)__hip_atomic_load(cfg.frowcnt + pil, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    double f[2 * K];
    {
      const double* F = cfg.fst + (size_t)pil * cfg.fstride + pr;
#pragma unroll
      for (int c = 0; c < 2 * K; ++c) f[c] = F[c * ks];
      // (the loads belong BEFORE the wait for the separator - the workgroup is idle there; left to itself the compiler
      // has moved them behind it in one build and not in the next: 2.7 against 9.5 us from the separator's flag to the
      // back substitution at K = 23)
#pragma unroll
      for (int c = 0; c < 2 * K; ++c) asm volatile("" : "+v"(f[c]));
    }
    double f2[2 * K];
    {
      const double* F = cfg.fst + (size_t)qil * cfg.fstride + qr;
#pragma unroll
      for (int c = 0; c < 2 * K; ++c) f2[c] = F[c * ks];
#pragma unroll
      for (int c = 0; c < 2 * K; ++c) asm volatile("" : "+v"(f2[c]));
    }
    if (tid == 0)
      spin_wait([&] { return __hip_atomic_load(cfg.sepflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch; }, cfg.spin);
    __syncthreads();
    (void)__hip_atomic_load(cfg.sepflag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    double* xs = Wm;   // (the augmented block is free between the passes)
    for (int c = tid; c < 2 * K; c += nt) {
      const int half = c / K, r = c - half * K;
      xs[c] = cfg.xsep[(mirror ? half : 1 - half) * K + r];   // mirrored chain: nearest = s, else nearest = s + 1
    }
    __syncthreads();
    if (tid < nloc * K) {
      double acc = 0.0;
#pragma unroll
      for (int c = 0; c < 2 * K; ++c) acc = __builtin_fma(f[c], xs[c], acc);
      lds[xall_off + (pil + 2) * ks + pr] -= acc;
    }
    if (tid + nt < nloc * K) {
      double acc = 0.0;
#pragma unroll
      for (int c = 0; c < 2 * K; ++c) acc = __builtin_fma(f2[c], xs[c], acc);
      lds[xall_off + (qil + 2) * ks + qr] -= acc;
    }
    for (int idx = tid + 2 * nt; idx < nloc * K; idx += nt) {   // (still more: the rest, unprefetched)
      const int il = idx / K, r = idx - il * K;
      const double* F = cfg.fst + (size_t)il * cfg.fstride + r;
      double acc = 0.0;
#pragma unroll
      for (int c = 0; c < 2 * K; ++c) acc = __builtin_fma(F[c * ks], xs[c], acc);
      lds[xall_off + (il + 2) * ks + r] -= acc;
    }
    __syncthreads();
  }
  chain_ts(cfg, 3);

  if (cfg.factor_only) return;
  // ---- backward pass: (D_i^-1 U_i) x_i = D_i^-1 rt_i - (D_i^-1 Ht_i) x_{i+1} - (D_i^-1 Et_i) x_{i+2}
  // One wavefront per right-hand side, no LDS staging and no barrier: lane = row r (+32 for the
  // second half).  The first half holds row r of D^-1 Ht_i, the second half row r of D^-1 Et_i,
  // both halves row r of the unit upper triangular D^-1 U_i (strict upper part; rows are
  // 16-byte loads straight from HBM, prefetched one block row ahead in a second register set).
  // x_{i+1}, x_{i+2} are wave-uniform (the back substitution produces each x_j by v_readlane).
  constexpr int KS2 = K * ks, KP = (K + 1) / 2;
  const int r_ = lane & 31, half_ = lane >> 5;
  const int rr = (r_ < K) ? r_ : 0;
  const bool live = r_ < K && half_ == 0;
  if (nrhs == 1 && bl_size) {
    // ---- single right-hand side: "push" form.  While the triangular solve of block row i yields
    // x_i one component per step (v_readlane + FMA, ~30 cycles of dependent latency), each
    // component is pushed at once into the pending right-hand sides of rows i-1 (through
    // D^-1 Ht_{i-1}) and i-2 (through D^-1 Et_{i-2}): no separate mat-vec phase, no wave-uniform
    // copy of x.  One accumulator per lane:
    //   lanes  0..31 (row r): acc = rhs of row i (the chain), FMA with D^-1 U_i;  w2 -= Et_{i-2}[r][j] x_j
    //   lanes 32..63 (row r): acc = rhs of row i-1,           FMA with D^-1 Ht_{i-1}
    // (the same instruction serves both halves), operands prefetched one block row ahead.
    if (wave == 0) {
      auto rowsA = [&](int i, double2 (&A)[KP]) {  // lower: U_i row r ; upper: Ht_{i-1} row r
        const int t = half_ ? i - 1 : i;
        if (t < 0) {
#pragma unroll
          for (int m = 0; m < KP; ++m) A[m] = make_double2(0.0, 0.0);
          return;
        }
        const double2* a = reinterpret_cast<const double2*>((half_ ? Hst : Ust) + (size_t)orig(t) * KS2 + rr * ks);
#pragma unroll
        for (int m = KP - 1; m >= 0; --m) A[m] = a[m];
      };
      auto rowsE = [&](int i, double2 (&E)[KP], double& w2init) {  // lower: Et_{i-2} row r and D^-1 rt_{i-2}
        const int t = i - 2;
        w2init = 0.0;
        if (t < 0 || half_) {
#pragma unroll
          for (int m = 0; m < KP; ++m) E[m] = make_double2(0.0, 0.0);
          return;
        }
        const double2* e = reinterpret_cast<const double2*>(Est + (size_t)orig(t) * KS2 + rr * ks);
#pragma unroll
        for (int m = KP - 1; m >= 0; --m) E[m] = e[m];
        w2init = Dst[(size_t)orig(t) * K + rr] * lds[xall_off + (t + 2) * ks + rr];
      };
      auto dvrt = [&](int t) {
        return (t < 0) ? 0.0 : Dst[(size_t)orig(t) * K + rr] * lds[xall_off + (t + 2) * ks + rr];
      };
      auto swap_halves = [&](double a) {  // value held by lane (l ^ 32)
        const unsigned lo = (unsigned)__double2loint(a), hi = (unsigned)__double2hiint(a);
        const auto slo = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto shi = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        return half_ ? __hiloint2double((int)shi[0], (int)slo[0]) : __hiloint2double((int)shi[1], (int)slo[1]);
      };
      double* xjoin = xch + 2 * (size_t)(K + ncr) * ks;  // [2][K]: x_m, x_{m+1}
      double acc = half_ ? dvrt(nloc - 2) : dvrt(nloc - 1);
      if (two && producer) {
        // x_{nloc} (= row m+1) and x_{nloc+1} (= row m) come from the top workgroup: their pushes
        if (lane == 0)
          spin_wait([&] { return __hip_atomic_load(flags + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch; }, cfg.spin);
        (void)__hip_atomic_load(flags + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        const double X0 = xjoin[K + rr], X1 = xjoin[rr];  // lane r: component r of x_{nloc}, x_{nloc+1}
        const double* h1 = Hst + (size_t)orig(nloc - 1) * KS2 + rr * ks;
        const double* e1 = Est + (size_t)orig(nloc - 1) * KS2 + rr * ks;
        const double* e2 = Est + (size_t)orig(nloc >= 2 ? nloc - 2 : 0) * KS2 + rr * ks;
        double vv = 0.0, ww = 0.0;
        for (int jj = 0; jj < K; ++jj) {
          const double x0 = rdlane(X0, jj), x1 = rdlane(X1, jj);
          vv = __builtin_fma(e1[jj], x1, vv);
          vv = __builtin_fma(h1[jj], x0, vv);
          ww = __builtin_fma(e2[jj], x0, ww);
        }
        acc -= half_ ? ((nloc >= 2) ? ww : 0.0) : vv;
      }
      double2 A0[KP], E0[KP], A1[KP], E1[KP];
      double wi0 = 0.0, wi1 = 0.0;
      auto solve = [&](int i, const double2 (&A)[KP], const double2 (&E)[KP], double w2) {
#pragma unroll
        for (int jj = K - 1; jj >= 0; --jj) {
          const double xj = rdlane(acc, jj);  // x_i[jj]: final once the steps above it are done
          const double ajj = (jj & 1) ? A[jj / 2].y : A[jj / 2].x;
          const double ejj = (jj & 1) ? E[jj / 2].y : E[jj / 2].x;
          acc = __builtin_fma(-ajj, xj, acc);  // (U strictly upper: rows >= jj keep their value)
          w2 = __builtin_fma(-ejj, xj, w2);
        }
        if (live) {
          lds[xall_off + (i + 2) * ks + r_] = acc;
          if (two && !producer && i >= m_split) xjoin[(size_t)(i - m_split) * K + r_] = acc;
        }
        if (two && !producer && i == m_split) {  // rows m+1 and m are solved: release the other workgroup
          __threadfence();
          if (lane == 0) __hip_atomic_store(flags + 1, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        // rotate: the lower half continues with the upper half's rhs (row i-1), the upper half
        // takes over the lower half's w2 (row i-2)
        acc = swap_halves(half_ ? acc : w2);
      };
      rowsA(nloc - 1, A0); rowsE(nloc - 1, E0, wi0);
      for (int i = nloc - 1; i >= 0; i -= 2) {
        rowsA(i - 1, A1); rowsE(i - 1, E1, wi1);
        solve(i, A0, E0, wi0);
        if (i - 1 >= 0) {
          rowsA(i - 2, A0); rowsE(i - 2, E0, wi0);
          solve(i - 1, A1, E1, wi1);
        }
      }
    }
  } else
  for (int j = wave; j < nrhs; j += NW) {
    double xs1[2 * KP], xs2[2 * KP];  // x_{i+1}, x_{i+2}, wave-uniform
#pragma unroll
    for (int c = 0; c < 2 * KP; ++c) { xs1[c] = 0.0; xs2[c] = 0.0; }
    double2 A0[KP], U0[KP], A1[KP], U1[KP];
    double dv0 = 1.0, dv1 = 1.0, rt0 = 0.0, rt1 = 0.0;
    double* xjoin = xch + 2 * (size_t)(K + ncr) * ks;  // [nrhs][2][K]: x_m, x_{m+1}
    auto load_row = [&](int i, double2 (&A)[KP], double2 (&U)[KP], double& dv, double& rtv) {
      if (i < 0) return;
      const size_t io = (size_t)orig(i);
      const double2* a = reinterpret_cast<const double2*>((half_ ? Est : Hst) + io * KS2 + rr * ks);
      const double2* u = reinterpret_cast<const double2*>(Ust + io * KS2 + rr * ks);
#pragma unroll
      for (int m = 0; m < KP; ++m) { A[m] = a[m]; U[m] = u[m]; }
      dv = Dst[io * K + rr];
      rtv = bl_size ? lds[xall_off + (j * (n + 2) + i + 2) * ks + rr]
                      : ((rr < k) ? x[(size_t)j * nk + io * k + rr] : 0.0);
    };
    // xa = x_{i+1}, xb = x_{i+2}; x_i overwrites xb (roles swap from row to row: no copies)
    auto solve_row = [&](int i, const double2 (&A)[KP], const double2 (&U)[KP], double dv, double rtv,
                         double (&xa)[2 * KP], double (&xb)[2 * KP]) {
      double acca = 0.0, accb = 0.0;  // both products per lane (uniform operands), one select after
#pragma unroll
      for (int m = 0; m < KP; ++m) {
        acca = __builtin_fma(A[m].x, xa[2 * m], acca);
        acca = __builtin_fma(A[m].y, xa[2 * m + 1], acca);
        accb = __builtin_fma(A[m].x, xb[2 * m], accb);
        accb = __builtin_fma(A[m].y, xb[2 * m + 1], accb);
      }
      double acc = half_ ? accb : acca;
      acc += __shfl_xor(acc, 32);
      double v = __builtin_fma(rtv, dv, -acc);
#pragma unroll
      for (int jj = K - 1; jj >= 0; --jj) {
        const double xj = rdlane(v, jj);
        xb[jj] = xj;
        const double ujj = (jj & 1) ? U[jj / 2].y : U[jj / 2].x;
        v = __builtin_fma(-ujj, xj, v);   // U is strictly upper: lanes >= jj keep their value
      }
      if (live) {
        if (bl_size) lds[xall_off + (j * (n + 2) + i + 2) * ks + r_] = v;
        else if (r_ < k) x[(size_t)j * nk + (size_t)orig(i) * k + r_] = v;
        // the join rows' solution is what the other workgroup's back substitution starts from
        if (two && !producer && i >= m_split) xjoin[(size_t)(j * 2 + (i - m_split)) * K + r_] = v;
      }
    };
    if (two && producer) {
      // x of local rows nloc (= row m+1) and nloc+1 (= row m) come from the top workgroup
      if (lane == 0)
        spin_wait([&] { return __hip_atomic_load(flags + 1 + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch; }, cfg.spin);
      (void)__hip_atomic_load(flags + 1 + j, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int c = 0; c < K; ++c) {
        xs1[c] = rdlane(xjoin[(size_t)(j * 2 + 1) * K + c], 0);  // wave-uniform: keep them in SGPRs
        xs2[c] = rdlane(xjoin[(size_t)(j * 2 + 0) * K + c], 0);
      }
    }
    load_row(nloc - 1, A0, U0, dv0, rt0);
    for (int i = nloc - 1; i >= 0; i -= 2) {
      load_row(i - 1, A1, U1, dv1, rt1);
      solve_row(i, A0, U0, dv0, rt0, xs1, xs2);       // x_i -> xs2
      if (i - 1 >= 0) {
        load_row(i - 2, A0, U0, dv0, rt0);
        solve_row(i - 1, A1, U1, dv1, rt1, xs2, xs1);  // x_{i-1} -> xs1 ; then xs1 = x_{i-1}, xs2 = x_i
      }
      if (two && !producer && i == nloc - 1) {  // rows m+1 and m are solved: release the other workgroup
        __threadfence();
        if (lane == 0) __hip_atomic_store(flags + 1 + j, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  __syncthreads();
  if (bl_size) {
    for (int idx = tid; idx < nrhs * nloc * k; idx += nt) {
      const int j = idx / (nloc * k), rem = idx - j * (nloc * k), i = rem / k, r = rem - i * k;
      x[(size_t)j * nk + (size_t)orig(i) * k + r] = lds[xall_off + (j * (n + 2) + i + 2) * ks + r];
    }
  }
  stamp(nfwd, 1);
  chain_ts(cfg, 4);
}

// K = compile-time block size >= k; the k x k blocks are embedded in K x K ones padded with
// the identity (padding rows/columns never mix with the real ones).
// b, x: [nrhs][n*k]; Ust/Hst/Est: [n][K*K] factors (internal layout), Dst: [n][K].
// RECT: the chain may be asked (cfg.rec_tail) for its back substitution in recursion form (the seven-workgroup kernel, K > 20)
template <int K, int NT, bool PADDED, int GJW, bool RECT = false>
__device__ __forceinline__ void
penta_ldl_body(int n, int k, const double* __restrict__ HA, const double* __restrict__ HB,
               const double* __restrict__ HC, const double* __restrict__ b, double rhs_sign, int nrhs,
               double* __restrict__ x, double* __restrict__ Ust, double* __restrict__ Hst,
               double* __restrict__ Est, double* __restrict__ Dst, double* __restrict__ dbg,
               const ChainCfg cfg, double* __restrict__ xch, unsigned* __restrict__ flags, unsigned epoch,
               unsigned* __restrict__ status, unsigned fact_id) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // ---- two-sided ("twisted") elimination, m_split > 0, grid of 2 workgroups: workgroup 0
  // eliminates block rows 0 .. m-1 top-down, workgroup 1 rows n-1 .. m+2 bottom-up with the
  // mirrored recursion (the reversed matrix has B'_{i'} = B_{n-i'}^T, A'_{i'} = A_{n+1-i'}^T: the
  // prefetch reads the bands transposed, nothing else changes).  Rows m, m+1 are the join: the
  // bottom workgroup runs two product-only pseudo-rows whose augmented blocks W are exactly its
  // Schur-complement contributions to rows m+1 and m, hands them over through `xch` + a flag,
  // and the top workgroup adds them before eliminating rows m and m+1.  Back substitution runs
  // outwards from the join in both workgroups (x_m, x_{m+1} handed to the bottom one).  The
  // dependent chain is ~n/2 block rows instead of n in both passes.
  const bool two = cfg.two != 0, mirror = cfg.mirror != 0, producer = cfg.producer != 0;
  const int m_split = cfg.m_split;
  const int nloc = cfg.nloc;                                          // block rows eliminated here
  const int nfwd = nloc + ((two && producer) ? 2 : 0);                // + pseudo-rows (producer)
  auto orig = [&](int il) { const int o = mirror ? cfg.base - il : cfg.base + il; return o < 0 ? 0 : o; };
  constexpr int nt = NT, KK = K * K, ks = ldl_ks(K), NW = NT / 64;
  const int kk = k * k;
  const int ncr = 2 * K + nrhs, per_wave = 64 - K;
  const int gj_waves = GJW ? GJW : (ncr + per_wave - 1) / per_wave;
  const size_t nk = (size_t)n * k;
  const PentaLdlLds L = penta_ldl_layout(n, K, nrhs, cfg.lds_rows);
  double* Wm = lds + L.W;
  auto stamp = [&](int i, int ph) {
    if (dbg && lane == 0)
      dbg[((cfg.dbg_slot * (NT / 64) + wave) * (n + 3) + i) * 8 + ph] = (double)__builtin_readcyclecounter();
  };

  chain_ts(cfg, 0);
  // ---- setup
  for (int idx = tid; idx < L.bl; idx += nt) lds[idx] = 0.0;
  for (int idx = L.xall + tid; idx < L.end; idx += nt) lds[idx] = 0.0;
  __syncthreads();
  for (int idx = tid; idx < 3 * ks; idx += nt) lds[L.Iv + idx] = ((idx % ks) < K) ? 1.0 : 0.0;
  for (int idx = tid; idx < L.bl_size; idx += nt) {  // layout [j][i][r] with K rows
    const int j = idx / (n * K), rem = idx - j * (n * K), i = rem / K, r = rem - i * K;
    lds[L.bl + idx] = (r < k && i < nloc) ? rhs_sign * b[(size_t)j * nk + (size_t)orig(i) * k + r] : 0.0;
  }

  // ---- row inputs [A_i | B_{i+1} | C_i | A_{i+2}]: fetched one row ahead into registers and
  // staged into LDS by the wavefronts that do NOT eliminate, during the elimination of the
  // previous row: the wavefront on the critical path never waits on vmcnt.  HA/HB/HC carry two
  // extra zero blocks, so rows i+1, i+2 need no guard.  Exact block size (k == K): the staged
  // layout equals the source layout; padded (k < K): identity padding.
  // helper wavefronts (those that do not eliminate): the first one forms G, the others stage /
  // prefetch / write back; with a single helper wavefront it does everything.
  const int nhelp = NW - gj_waves;
  const bool g_wave = wave == gj_waves;                         // forms G
  // (with only two helpers the one that forms G shares the I/O: a lone I/O wavefront is the
  // bottleneck of the K = 23 / 24 instantiations, 10.6k cycles per row against 3.4k for G)
  const int io_first = (nhelp > 2) ? gj_waves + 1 : gj_waves;   // first wavefront doing the I/O
  const int ht = tid - io_first * 64;                           // index among the I/O threads (< 0: none)
  const int hn = nt - io_first * 64;
  constexpr int HN_MIN = GJW ? ((NT / 64 - GJW > 2) ? NT - (GJW + 1) * 64 : NT - GJW * 64) : 64;
  constexpr int PMAX = (3 * KK + HN_MIN - 1) / HN_MIN;
  // Branch-free: every I/O lane loads PMAX values per row through 32-bit offsets from HA (the
  // three bands live in one allocation); lanes/slots without a source load element 0 and the
  // result is discarded (m_valid) or replaced by the identity padding (m_load / m_one).
  double pre[PMAX];
  int p_off[PMAX];
  unsigned p_col[(PMAX + 3) / 4] = {0};  // staged column (8 bits per slot) of the A-block slots
  unsigned long long m_valid = 0, m_load = 0, m_one = 0, m_isA = 0, m_isB = 0;
  static_assert(PMAX <= 64, "slot masks are 64-bit");
  const int dHB = (int)(HB - HA), dHC = (int)(HC - HA);
#pragma unroll
  for (int s = 0; s < PMAX; ++s) {
    const int idx = ht + s * hn;
    p_off[s] = 0;
    if (ht >= 0 && idx < 3 * KK) {  // staged blocks: 0 B_{i+1}, 1 C_i, 2 A_{i+2} (top-down naming)
      m_valid |= 1ull << s;
      const int which = idx / KK, e = idx - which * KK, c = e / K, r = e - c * K;
      if (which == 0) m_isB |= 1ull << s;
      if (which == 2) { m_isA |= 1ull << s; p_col[s / 4] |= c << (8 * (s & 3)); }
      if (!PADDED || (r < k && c < k)) {
        m_load |= 1ull << s;
        if (which == 1) p_off[s] = dHC + c * k + r;
        else if (!mirror) p_off[s] = ((which == 0) ? dHB + kk : 2 * kk) + c * k + r;  // B_{i+1}, A_{i+2}
        else p_off[s] = ((which == 0) ? dHB : 0) + r * k + c;                        // B_i^T, A_i^T
      } else if (which == 1 && r == c) {
        m_one |= 1ull << s;
      }
    }
  }
  auto fetch = [&](int i) {
    const double* base = HA + (size_t)orig(i) * kk;
#pragma unroll
    for (int s = 0; s < PMAX; ++s) pre[s] = base[p_off[s]];
  };
  auto stage = [&](int il) {  // il = the local row being staged
    // join rows of the top workgroup have no coupling to rows m+2.. (eliminated by the other
    // workgroup); pseudo-rows of the bottom workgroup have all-zero inputs
    unsigned long long kill = 0;
    if (two) {
      if (!producer) kill = (il >= m_split ? m_isA : 0ull) | (il >= m_split + 1 ? m_isB : 0ull);
      else if (il >= nloc) kill = ~0ull;
    }
#pragma unroll
    for (int s = 0; s < PMAX; ++s) {
      double val = pre[s];
      if (PADDED) val = (m_load >> s & 1) ? val : ((m_one >> s & 1) ? 1.0 : 0.0);
      val = (kill >> s & 1) ? 0.0 : val;
      // B and C keep the source layout; A_{i+2} goes transposed into the E columns' layout
      // (element (row r, col c) of the staged block is E(c, r)): Eb[r * ks + c]
      int dst = L.in + ht + s * hn;
      if (m_isA >> s & 1) {
        const int e = ht + s * hn - 2 * KK, c = (p_col[s / 4] >> (8 * (s & 3))) & 0xff;
        dst = L.Eb + (il & 1) * L.kks + e * ks - c * (K * ks - 1);
      }
      dst = (m_valid >> s & 1) ? dst : L.dump;
      lds[dst] = val;
    }
  };
  // write-back jobs (fixed over rows): element idx = r*ks + c of the row-major factor blocks;
  // packed: source offset c*ks + r | r << 16 | (c >= K) << 24 | (r < c < K) << 25
  constexpr int WBMAX = (K * ks + HN_MIN - 1) / HN_MIN;
  int wb_src[WBMAX];
#pragma unroll
  for (int it = 0; it < WBMAX; ++it) {
    const int idx = (ht >= 0 ? ht : 0) + it * hn;
    const int r = (idx / ks < K) ? idx / ks : 0, c = idx - (idx / ks) * ks;
    const int cc = (c < K) ? c : 0;
    wb_src[it] = (cc * ks + r) | r << 16 | (c >= K ? 1 : 0) << 24 | ((r < c && c < K) ? 1 : 0) << 25;
  }
  if (ht >= 0) { fetch(0); stage(0); fetch(1); }

  // ---- block products on the matrix cores: v_mfma_f64_16x16x4 computes a 16x16 tile of
  // X^T diag(dn) Y per SK = ceil(K/4) instructions.  Operand layout (one f64 per lane):
  //   A[i][kq] = X(4s + kq, 16 tr + i),  B[kq][j] = dn(4s + kq) Y(4s + kq, 16 tc + j),
  //   i, j = lane & 15, kq = lane >> 4;  D[(lane >> 4) + 4 reg][lane & 15], reg = 0..3.
  // Blocks are K-row columns with stride ks in LDS, so A and B fragments are both
  // X[(16 t + (lane & 15)) * ks + 4 s + (lane >> 4)]: one ds_read_b64 per (tile index, k-step).
  // Columns >= K of a tile read whatever follows the block in LDS; they only feed outputs that
  // are never stored.  Rows >= K (the zero pad of every column, and dn = 0 there) add nothing.
  using d4 = __attribute__((ext_vector_type(4))) double;
  constexpr int SK = (K + 3) / 4;    // k-steps per tile
  constexpr int TT = (K + 15) / 16;  // 16-wide tiles per dimension (1 or 2: K <= 32)
  static_assert(4 * SK <= ks, "k-steps run into the next column");
  const int fl = lane & 15, fk = lane >> 4;
  auto ldop = [&](const double* Xp, int t, double (&a)[SK]) {
    const double* pp = Xp + (16 * t + fl) * ks + fk;
#pragma unroll
    for (int sq = 0; sq < SK; ++sq) a[sq] = pp[4 * sq];
  };
  auto mma = [&](const double (&a)[SK], const double (&bq)[SK], const double (&dnv)[SK], d4& acc) {
#pragma unroll
    for (int sq = 0; sq < SK; ++sq) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[sq], bq[sq] * dnv[sq], acc, 0, 0, 0);
  };
  const bool ypush = (nrhs == 1 && gj_waves == 1 && 3 * K < 64);
  __syncthreads();

  constexpr int XPRE = RECT ? (K * ldl_ks(K) + NT - 1) / NT : 1;
  double xpre[XPRE] = {}, ypre = 0.0;   // (RECT: the second join row's contributions, fetched during the first)
  for (int i = 0; i < nfwd; ++i) {
    const bool pseudo = i >= nloc;  // bottom workgroup: product-only rows forming the join contributions
    double* Bn = lds + L.in;
    double* Ci = Bn + KK;
    double* An2 = Ci + KK;
    const double* Htp = lds + L.Ht + ((i + 1) & 1) * L.kks;     // Ht_{i-1}
    double* Htn = lds + L.Ht + (i & 1) * L.kks;                 // Ht_i
    const double* Etp = lds + L.Et + ((i + 2) % 3) * L.kks;     // Et_{i-1}
    const double* Etpp = lds + L.Et + ((i + 1) % 3) * L.kks;    // Et_{i-2}
    double* Etn = lds + L.Et + (i % 3) * L.kks;                 // Et_i
    const double* Ivp = lds + L.Iv + ((i + 2) % 3) * ks;        // Dn_{i-1}
    const double* Ivpp = lds + L.Iv + ((i + 1) % 3) * ks;       // Dn_{i-2}
    double* Ivn = lds + L.Iv + (i % 3) * ks;
    // rt_i of right-hand side j: ring slot, or row i+2 of xall (then the ring is not used)
    const int rtj = L.bl_size ? (n + 2) * ks : ks;  // stride between right-hand sides
    const double* rtp = L.bl_size ? lds + L.xall + (i + 1) * ks : lds + L.rt + ((i + 2) % 3) * L.rts;
    const double* rtpp = L.bl_size ? lds + L.xall + i * ks : lds + L.rt + ((i + 1) % 3) * L.rts;
    double* rtn = L.bl_size ? lds + L.xall + (i + 2) * ks : lds + L.rt + (i % 3) * L.rts;
    double* Un = lds + L.U + (i & 1) * L.kks;
    const double* Up = lds + L.U + ((i + 1) & 1) * L.kks;
    double* Gb = lds + L.G;

    stamp(i, 0);
    stamp(i, 7);
    stamp(i, 1);

    // ---- block products (MFMA); tiles per wavefront, TT = 2:  0: S(0,0)          1: S(1,0) H(1,0)
    //                                                            2: H(0,0) S(1,1)   3: H(0,1) H(1,1)
    //                                                  TT = 1:  0: S(0,0)          1: H(0,0)
    {
      // epilogues: all LDS reads first, then all writes (the compiler cannot move a read above a
      // write to the same address space, and a lone wavefront pays the full LDS latency per
      // dependent read -> write pair)
      auto put_s = [&](int tr, int tc, const d4& acc) {  // S = C - G - Ht^T Dn Ht (lower part, mirrored)
        double cv[4], gv[4];
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int r = 16 * tr + fk + 4 * rg, c = 16 * tc + fl;
          const bool on = r < K && c < K && r >= c;
          cv[rg] = on ? Ci[c * K + r] : 0.0;
          gv[rg] = on ? Gb[c * K + r] : 0.0;
        }
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int r = 16 * tr + fk + 4 * rg, c = 16 * tc + fl;
          if (r < K && c < K && r >= c) {
            const double val = (cv[rg] - gv[rg]) - acc[rg];
            Wm[c * ks + r] = val;
            if (r != c) Wm[r * ks + c] = val;
          }
        }
      };
      auto put_h = [&](int tr, int tc, const d4& acc) {  // H = B_{i+1}^T - Ht^T Dn Et_{i-1}
        double bv[4];
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int r = 16 * tr + fk + 4 * rg, c = 16 * tc + fl;
          bv[rg] = (r < K && c < K) ? Bn[r * K + c] : 0.0;
        }
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int r = 16 * tr + fk + 4 * rg, c = 16 * tc + fl;
          if (r < K && c < K) Wm[(K + c) * ks + r] = bv[rg] - acc[rg];
        }
      };
      double dnv[SK];
      if (wave < 4) {
#pragma unroll
        for (int sq = 0; sq < SK; ++sq) dnv[sq] = Ivp[4 * sq + fk];
      }
      const d4 zero4 = {0.0, 0.0, 0.0, 0.0};
      if (TT == 1) {
        if (wave == 0) {
          double a0[SK]; ldop(Htp, 0, a0);
          d4 acc = zero4; mma(a0, a0, dnv, acc); put_s(0, 0, acc);
        } else if (wave == 1) {
          double a0[SK], e0[SK]; ldop(Htp, 0, a0); ldop(Etp, 0, e0);
          d4 acc = zero4; mma(a0, e0, dnv, acc); put_h(0, 0, acc);
        }
      } else {
        if (wave == 0) {  // the wavefront that eliminates next gets the least
          double a0[SK]; ldop(Htp, 0, a0);
          d4 s00 = zero4;
          mma(a0, a0, dnv, s00);
          put_s(0, 0, s00);
        } else if (wave == 1) {
          double a0[SK], a1[SK], e0[SK]; ldop(Htp, 1, a1); ldop(Htp, 0, a0); ldop(Etp, 0, e0);
          d4 s10 = zero4, h10 = zero4;
          mma(a1, a0, dnv, s10); mma(a1, e0, dnv, h10);
          put_s(1, 0, s10); put_h(1, 0, h10);
        } else if (wave == 2) {
          double a0[SK], a1[SK], e0[SK]; ldop(Htp, 0, a0); ldop(Etp, 0, e0); ldop(Htp, 1, a1);
          d4 h00 = zero4, s11 = zero4;
          mma(a0, e0, dnv, h00); mma(a1, a1, dnv, s11);
          put_h(0, 0, h00); put_s(1, 1, s11);
        } else if (wave == 3) {
          double a0[SK], a1[SK], e1[SK]; ldop(Htp, 0, a0); ldop(Htp, 1, a1); ldop(Etp, 1, e1);
          d4 h01 = zero4, h11 = zero4;
          mma(a0, e1, dnv, h01); mma(a1, e1, dnv, h11);
          put_h(0, 1, h01); put_h(1, 1, h11);
        }
      }
    }
    // ---- right-hand side column
    if (ypush) {
      // y = r - (Ht_{i-1}^T Dn rt_{i-1}) - (Et_{i-2}^T Dn rt_{i-2}): both products were accumulated by
      // the elimination wavefront of the previous two rows (y-push), only the subtraction is left
      if (wave == 2 && lane < K) {
        const double bval = L.bl_size ? lds[L.bl + i * K + lane]
                                      : ((lane < k && !pseudo) ? rhs_sign * b[(size_t)orig(i) * k + lane] : 0.0);
        Wm[3 * K * ks + lane] = (bval - lds[L.yh + ((i + 1) & 1) * ks + lane]) - lds[L.ye + ((i + 1) % 3) * ks + lane];
      }
    } else if (gj_waves >= 2 && NW >= 4 && nrhs == 1 && K <= 32) {
      // two elimination wavefronts, one right-hand side: wavefront 0 (one product tile only in this
      // phase) forms y with the two mat-vecs on its two halves - lanes 0..31 (Ht_{i-1}^T Dn rt_{i-1})[r],
      // lanes 32..63 (Et_{i-2}^T Dn rt_{i-2})[r] - and one cross-half exchange
      if (wave == 0) {
        const int r = lane & 31, hf = lane >> 5, rr2 = (r < K) ? r : 0;
        const double2* m2 = reinterpret_cast<const double2*>((hf ? Etpp : Htp) + rr2 * ks);
        const double2* dd2 = reinterpret_cast<const double2*>(hf ? Ivpp : Ivp);
        const double2* rv2 = reinterpret_cast<const double2*>(hf ? rtpp : rtp);
        double acc = 0;
#pragma unroll
        for (int m = 0; m < (K + 1) / 2; ++m) {
          const double2 mm = m2[m], da = dd2[m], ra = rv2[m];
          acc = __builtin_fma(mm.x * da.x, ra.x, acc);
          acc = __builtin_fma(mm.y * da.y, ra.y, acc);
        }
        const double other = __shfl_xor(acc, 32);
        if (hf == 0 && r < K) {
          const double bval = L.bl_size ? lds[L.bl + i * K + r]
                                        : ((r < k && !pseudo) ? rhs_sign * b[(size_t)orig(i) * k + r] : 0.0);
          Wm[3 * K * ks + r] = (bval - acc) - other;
        }
      }
    } else if ((gj_waves >= 2 && NW >= 4) ? (wave < 2) : (wave >= 2 || NW < 3)) {
      // one thread per (rhs, row), spread over the wavefronts >= 2 - or, with two elimination
      // wavefronts, over those two: they have the fewest product tiles in this phase
      const bool first2 = gj_waves >= 2 && NW >= 4;
      const int w2 = (NW < 3 || first2) ? wave : wave - 2, nw2 = first2 ? 2 : ((NW < 3) ? NW : NW - 2);
      for (int t = w2 * 64 + lane; t < nrhs * K; t += nw2 * 64) {
        const int j = t / K, r = t - j * K;
        double a0 = 0, a1 = 0;
        {
          const double2* h2 = reinterpret_cast<const double2*>(Htp + r * ks);
          const double2* e2 = reinterpret_cast<const double2*>(Etpp + r * ks);
          const double2* d1 = reinterpret_cast<const double2*>(Ivp);
          const double2* d2 = reinterpret_cast<const double2*>(Ivpp);
          const double2* r1 = reinterpret_cast<const double2*>(rtp + j * rtj);
          const double2* r2 = reinterpret_cast<const double2*>(rtpp + j * rtj);
#pragma unroll
          for (int m = 0; m < (K + 1) / 2; ++m) {
            const double2 hh = h2[m], ee = e2[m], da = d1[m], db = d2[m], ra = r1[m], rb = r2[m];
            a0 = __builtin_fma(hh.x * da.x, ra.x, a0); a0 = __builtin_fma(hh.y * da.y, ra.y, a0);
            a1 = __builtin_fma(ee.x * db.x, rb.x, a1); a1 = __builtin_fma(ee.y * db.y, rb.y, a1);
          }
        }
        const double bval = L.bl_size ? lds[L.bl + (j * n + i) * K + r]
                                      : ((r < k && !pseudo) ? rhs_sign * b[(size_t)j * nk + (size_t)orig(i) * k + r] : 0.0);
        Wm[(3 * K + j) * ks + r] = (bval - a0) - a1;
      }
    }
    stamp(i, 5);
    lds_barrier();
    stamp(i, 2);

    if (two && !producer && i >= m_split) {
      // ---- join: add the other workgroup's Schur-complement contributions to [S | H | . | y]
      if (i == m_split) {
        chain_ts(cfg, 1);
        if (tid == 0)
          spin_wait([&] { return __hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch; }, cfg.spin);
        __syncthreads();
        (void)__hip_atomic_load(flags, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);  // every wavefront acquires
        chain_ts(cfg, 5);
      }
      const int wsz = (K + ncr) * ks;
      const double* X0 = xch;         // pseudo-row 0: contributions to row m+1 (and the (m+1, m) coupling)
      const double* X1 = xch + wsz;   // pseudo-row 1: contributions to row m
      const double* Xs = (i == m_split) ? X1 : X0;
      if (RECT && nrhs == 1) {
        // (the second join row's contributions are asked for a row ahead: ~1.5 us of exposed load latency of that row)
        if (i == m_split) {
#pragma unroll
          for (int j = 0; j < XPRE; ++j) xpre[j] = X0[(tid + j * nt < K * ks) ? tid + j * nt : 0];
          ypre = X0[3 * K * ks + (tid < ks ? tid : 0)];
          for (int idx = tid; idx < K * ks; idx += nt) Wm[idx] += Xs[idx];
          if (tid < ks) Wm[3 * K * ks + tid] += Xs[3 * K * ks + tid];
        } else {
#pragma unroll
          for (int j = 0; j < XPRE; ++j) if (tid + j * nt < K * ks) Wm[tid + j * nt] += xpre[j];
          if (tid < ks) Wm[3 * K * ks + tid] += ypre;
        }
      } else {
        for (int idx = tid; idx < K * ks; idx += nt) Wm[idx] += Xs[idx];                       // S
        for (int idx = tid; idx < nrhs * ks; idx += nt) Wm[3 * K * ks + idx] += Xs[3 * K * ks + idx];  // y
      }
      if (i == m_split)
        for (int idx = tid; idx < KK; idx += nt) {  // H(r, c) += H'(c, r)
          const int c = idx / K, r = idx - c * K;
          Wm[(K + c) * ks + r] += X0[(K + r) * ks + c];
        }
      lds_barrier();
    }

    if (wave < gj_waves) {
      // ---- forward elimination in registers
      const int rc = wave * per_wave + (lane - K);
      const bool is_rhs = lane >= K && rc < ncr;
      const int col = (lane < K) ? lane : (is_rhs ? K + rc : 0);
      double xr[K];
      {
        // columns of [S | H | . | y] from W, the E columns straight from their staging buffer
        const double* colp = (col >= 2 * K && col < 3 * K) ? lds + L.Eb + (i & 1) * L.kks + (col - 2 * K) * ks : Wm + col * ks;
        const double2* src = reinterpret_cast<const double2*>(colp);
#pragma unroll
        for (int r2 = 0; r2 < K / 2; ++r2) { const double2 t2 = src[r2]; xr[2 * r2] = t2.x; xr[2 * r2 + 1] = t2.y; }
        if (K & 1) xr[K - 1] = colp[K - 1];
      }
      stamp(i, 3);
      double myinv = 1.0, yacc = 0.0;
      if (!pseudo) {
        // the diagonal entry this lane's pivot starts from (before the block's own elimination)
        const double diag0 = (lane < K) ? Wm[lane * ks + lane] : 1.0;
        if (ypush) myinv = ldl_eliminate_wave<K, true>(xr, lane, yacc);
        else myinv = ldl_eliminate_wave<K, false>(xr, lane, yacc);
        // factorisation status (the reference reports kFailure from Factorize and the optimizer
        // demands success: penta_diagonal_solver.h:181-185, trajectory_optimizer.cc:2084): a pivot
        // that is not positive, not finite, or has lost every significant digit against the diagonal
        // entry it started from (d <= eps * S_ll) means H is not numerically positive definite.
        // myinv = 1 / d from v_rcp + Newton: NaN for d = 0 / inf / NaN, negative for d < 0.
        if (wave == 0) {
          const bool bad = ldl_pivot_bad(myinv, diag0, lane, k, cfg.npos);
          if (__builtin_amdgcn_ballot_w64(bad) != 0ull && lane == 0) {
            __hip_atomic_store(status, fact_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_fetch_add(status + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (cfg.ts) cfg.ts[20] = (double)(i + 1);   // (debug: first/last failing local row of this chain)
          }
        }
      } else {
        // hand the column over (layout of W) and continue the recursion with a zero row
        if (lane < K || is_rhs) {
          double* dstg = xch + (size_t)(i - nloc) * (K + ncr) * ks + (size_t)col * ks;
#pragma unroll
          for (int r = 0; r < K; ++r) dstg[r] = xr[r];
        }
#pragma unroll
        for (int r = 0; r < K; ++r) xr[r] = 0.0;
      }
      stamp(i, 4);
      // every lane stores its column: S lanes -> U, rhs lanes -> Ht | Et | rt
      double* dst = nullptr;
      if (lane < K) dst = (wave == 0) ? Un + lane * ks : nullptr;
      else if (is_rhs) dst = (rc < K) ? Htn + rc * ks : ((rc < 2 * K) ? Etn + (rc - K) * ks : rtn + (rc - 2 * K) * rtj);
      if (dst) {
        double2* d2 = reinterpret_cast<double2*>(dst);
#pragma unroll
        for (int r2 = 0; r2 < K / 2; ++r2) d2[r2] = make_double2(xr[2 * r2], xr[2 * r2 + 1]);
        if (K & 1) dst[K - 1] = xr[K - 1];
      }
      if (wave == 0 && lane < K) Ivn[lane] = myinv;
      if (ypush) {  // contributions of this row to the right-hand sides of rows i+1 (lanes K..2K-1), i+2
        if (lane >= K && lane < 2 * K) lds[L.yh + (i & 1) * ks + lane - K] = yacc;
        else if (lane >= 2 * K && lane < 3 * K) lds[L.ye + (i % 3) * ks + lane - 2 * K] = yacc;
      }
      if (!L.bl_size && is_rhs && rc >= 2 * K) {
        const int j = rc - 2 * K;
#pragma unroll
        for (int r = 0; r < K; ++r)
          if (r < k && !pseudo) x[(size_t)j * nk + (size_t)orig(i) * k + r] = xr[r];  // parked until the backward pass
      }
    } else {
      // ---- helper wavefronts
      if (ht >= 0) {  // stage the next row's inputs, prefetch the one after
        stage(i + 1);
        stamp(i, 3);
        fetch(i + 2);
        stamp(i, 4);
      }
      if (g_wave) {   // G = Et_{i-1}^T Dn_{i-1} Et_{i-1} for the next row (symmetric), MFMA
        double dnv[SK], e0[SK];
#pragma unroll
        for (int sq = 0; sq < SK; ++sq) dnv[sq] = Ivp[4 * sq + fk];
        ldop(Etp, 0, e0);
        auto put_g = [&](int tr, int tc, const d4& acc) {
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int r = 16 * tr + fk + 4 * rg, c = 16 * tc + fl;
            if (r < K && c < K && r >= c) {
              Gb[c * K + r] = acc[rg];
              if (r != c) Gb[r * K + c] = acc[rg];
            }
          }
        };
        const d4 zero4 = {0.0, 0.0, 0.0, 0.0};
        d4 g00 = zero4;
        mma(e0, e0, dnv, g00);
        put_g(0, 0, g00);
        if (TT == 2) {
          double e1[SK];
          ldop(Etp, 1, e1);
          d4 g10 = zero4, g11 = zero4;
          mma(e1, e0, dnv, g10); mma(e1, e1, dnv, g11);
          put_g(1, 0, g10); put_g(1, 1, g11);
        }
      }
      if (ht >= 0 && i > 0 && i <= nloc) {
        // factors of row i-1 for the backward pass, ROW-major (stride ks) so that a lane reads its
        // row with 16-byte loads; rows are scaled by 1/d_r, U keeps its strict upper triangle
        if (cfg.rowcnt) {   // (a joiner of the seven-workgroup kernel: write-through, see below)
#pragma unroll
          for (int it = 0; it < WBMAX; ++it) {
            const int idx = ht + it * hn;
            if (idx < K * ks) {
              const int src = wb_src[it] & 0xffff, r = wb_src[it] >> 16 & 0xff;
              const bool in = !(wb_src[it] >> 24 & 1), up = wb_src[it] >> 25 & 1;
              const double dr = Ivp[r];
              const size_t at = (size_t)orig(i - 1) * K * ks + idx;
              __hip_atomic_store(Ust + at, up ? Up[src] * dr : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_store(Hst + at, in ? Htp[src] * dr : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_store(Est + at, in ? Etp[src] * dr : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
        } else {
#pragma unroll
          for (int it = 0; it < WBMAX; ++it) {
            const int idx = ht + it * hn;
            if (idx < K * ks) {
              const int src = wb_src[it] & 0xffff, r = wb_src[it] >> 16 & 0xff;
              const bool in = !(wb_src[it] >> 24 & 1), up = wb_src[it] >> 25 & 1;
              const double dr = Ivp[r];
              Ust[(size_t)orig(i - 1) * K * ks + idx] = up ? Up[src] * dr : 0.0;
              Hst[(size_t)orig(i - 1) * K * ks + idx] = in ? Htp[src] * dr : 0.0;
              Est[(size_t)orig(i - 1) * K * ks + idx] = in ? Etp[src] * dr : 0.0;
            }
          }
        }
        if (!cfg.rowcnt)
          for (int r = ht; r < K; r += hn) Dst[(size_t)orig(i - 1) * K + r] = Ivp[r];
        if (cfg.rowcnt) {
          // nested dissection: row i-1 (factors above, 1 / d and rt here) is complete in HBM once every I/O
          // wavefront's stores are acknowledged.  They are write-through stores: nothing stays dirty in this
          // XCD's L2, so the release is a wait for the acknowledgements instead of a write-back of that L2
          // (a releasing fence per I/O wavefront and row from each of 64 joiners cost a batch of 32 problems).
          // Ordered by the hardware, not by the HIP memory model: see penta_nd.h release_row for the assumption
          // (gfx942 / gfx950: agent-scope stores write through, vmcnt counts their acknowledgements).
          for (int r = ht; r < K; r += hn) {
            __hip_atomic_store(Dst + (size_t)orig(i - 1) * K + r, Ivp[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(cfg.rtpub + (size_t)(i - 1) * K + r, lds[L.xall + (i - 1 + 2) * ks + r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (lane == 0)
            __hip_atomic_fetch_add(cfg.rowcnt + (i - 1), cfg.rowcnt_unit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (cfg.ts && lane == 0 && i - 1 < 20) cfg.ts[24 + (i - 1)] = (double)wall_clock64();   // (the last wavefront's stamp stays: the row is out)
        }
      }
    }
    stamp(i, 6);
    lds_barrier();
  }
  if (nfwd == nloc) {  // last row's factors (the bottom workgroup wrote them during its pseudo-rows)
    const int i = nloc - 1;
    const double* Ul = lds + L.U + (i & 1) * L.kks;
    const double* Hl = lds + L.Ht + (i & 1) * L.kks;
    const double* El = lds + L.Et + (i % 3) * L.kks;
    const double* Il = lds + L.Iv + (i % 3) * ks;
    for (int idx = tid; idx < K * ks; idx += nt) {
      const int r = idx / ks, c = idx - r * ks;
      const bool in = c < K;
      const double dr = Il[r];
      Ust[(size_t)orig(i) * K * ks + idx] = (in && r < c) ? Ul[c * ks + r] * dr : 0.0;
      Hst[(size_t)orig(i) * K * ks + idx] = in ? Hl[c * ks + r] * dr : 0.0;
      Est[(size_t)orig(i) * K * ks + idx] = in ? El[c * ks + r] * dr : 0.0;
    }
    for (int r = tid; r < K; r += nt) Dst[(size_t)orig(i) * K + r] = Il[r];
    if (cfg.rowcnt)
      for (int r = tid; r < K; r += nt) cfg.rtpub[(size_t)i * K + r] = lds[L.xall + (i + 2) * ks + r];
  }
  if (two && producer) {  // publish the join contributions
    __threadfence();
    __syncthreads();
    if (tid == 0) __hip_atomic_store(flags, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();  // factors written by this block are re-read below: drain the stores
  __threadfence_block();
  if (cfg.rowcnt && tid == 0) {  // the last row was written by every wavefront (above): one release for all
    const int nio = (NT - io_first * 64) / 64;
    __hip_atomic_fetch_add(cfg.rowcnt + (nloc - 1), cfg.rowcnt_unit * nio, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
  stamp(nfwd, 0);
  chain_ts(cfg, 2);
  if constexpr (RECT && NT == 256) {
    if (cfg.rec_tail) { chain_recursion_tail<K>(n, k, x, Ust, Hst, Est, Dst, cfg, epoch, L.xall); return; }
  }
  penta_ldl_tail<K, NT>(n, k, nrhs, x, Ust, Hst, Est, Dst, dbg, cfg, xch, flags, epoch, L.xall, L.bl_size, L.W, nfwd);
}

// local rows (incl. the producer's two pseudo-rows, + 2 spare) either workgroup of the two-sided kernel touches
__host__ __device__ inline int ldl_two_sided_rows(int n, int m_split, int nrhs) {
  if (m_split <= 0 || nrhs != 1) return 0;
  const int top = m_split + 2, bottom = n - m_split;
  return (top > bottom ? top : bottom) + 2;
}
// the two roles of the two-workgroup kernel (m_split = 0: one workgroup, the whole system)
__host__ __device__ inline ChainCfg two_sided_cfg(int n, int m_split, int side) {
  ChainCfg c = {};
  c.two = m_split > 0;
  c.mirror = side; c.producer = side;
  c.base = side ? n - 1 : 0;
  c.nloc = c.two ? (side ? n - m_split - 2 : m_split + 2) : n;
  c.m_split = m_split;
  c.dbg_slot = side;
  return c;
}

// grid (sides, batch): blockIdx.x = side of the two-sided elimination, blockIdx.y = problem of the
// batch (arenas `pstride` bytes apart, see kernels.h at_problem; b / x are per-problem arrays too)
template <int K, int NT, bool PADDED, int GJW>
__global__ void __launch_bounds__(NT)
penta_ldl_kernel(int n, int k, const double* __restrict__ HA, const double* __restrict__ HB,
                 const double* __restrict__ HC, const double* __restrict__ b, double rhs_sign, int nrhs,
                 double* __restrict__ x, double* __restrict__ Ust, double* __restrict__ Hst,
                 double* __restrict__ Est, double* __restrict__ Dst, double* __restrict__ dbg,
                 int m_split, double* __restrict__ xch, unsigned* __restrict__ flags, unsigned epoch,
                 unsigned* __restrict__ status, unsigned fact_id, size_t pstride, int factor_only, int npos) {
  const size_t o = (size_t)blockIdx.y * pstride;
  ChainCfg cfg = two_sided_cfg(n, m_split, (int)blockIdx.x);
  cfg.factor_only = factor_only;
  cfg.npos = npos;
  cfg.lds_rows = ldl_two_sided_rows(n, m_split, nrhs);
  cfg.spin = SpinCtl{status + 2 * gridDim.y, fact_id};   // (behind the per-problem status words)
  penta_ldl_body<K, NT, PADDED, GJW>(n, k, at_problem(HA, o), at_problem(HB, o), at_problem(HC, o), at_problem(b, o),
                                     rhs_sign, nrhs, at_problem(x, o), at_problem(Ust, o), at_problem(Hst, o),
                                     at_problem(Est, o), at_problem(Dst, o), dbg ? at_problem(dbg, o) : nullptr,
                                     cfg, at_problem(xch, o), at_problem(flags, o),
                                     epoch, status + 2 * blockIdx.y, fact_id);
}

}  // namespace idto_dev
