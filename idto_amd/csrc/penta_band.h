// penta_band.h — the Gauss-Newton system of the small models (blocks of 2 .. 5: acrobot, spinner, hopper and their KKT
// systems) as what it also is: a SCALAR symmetric band matrix of half width 3 K - 1, factorised L D L^T pivot by pivot
// in the registers of one wavefront per chain (reference recursion: optimizer/penta_diagonal_solver.h:124-248).
//
// The block kernels (penta_ldl.h, penta_pipe.h) pay ~1.5 us per BLOCK row whatever K is - hand-overs between
// wavefronts, a products phase on the matrix cores, barriers: 35 us for acrobot's 41 rows of 2 x 2, most of its step
// (DESIGN.md 5.10, 5.11).  Here a pivot is ~20 + 6 W instructions of one wavefront and nothing else:
//
//   storage   column q of the lower band in 16 LDS cells: cell d = A[q + d][q] (d <= w = W - 1), cell 15 = the
//             right-hand side's b_q (the matrix bordered by b as its last row), the rest zero
//   window    the W = 3 K columns j .. j + W - 1, column q in register slot q mod W, lane d of a row of 16 lanes
//             holding cell d (the four rows of the wavefront hold copies)
//   pivot j   col = slot j mod W;  1 / d_j from lane 0;  l = col / d_j (lane 15: y_j / d_j);  for c = 1 .. w:
//                 slot (j + c) mod W  -=  (l shifted down by c lanes: v_mov_dpp row_shl; lane 15 stays) * A[j + c][j] (v_readlane)
//             = A[j + c + d][j + c] -= l_{c + d} A[j + c][j]  and  y_{j + c} -= (y_j / d_j) A[j + c][j];  l goes back
//             to the column's cells (the factor; cell 15: y_j / d_j), the slot is refilled with column j + W.
//             Groups of W pivots are straight-line code: every register index, lane index, shift and LDS offset is
//             an immediate, nothing is masked.
//   back      x_j = y_j / d_j + p_0;  p_d -= l_{j, j - d} x_j for the w rows above (row j of L: cells (j - d, d), one
//             LDS read that brings y_j / d_j along in lane 15), p shifted by one lane per step
//
// Two wavefronts take the two ends, each on a copy of ITS part of the band - the second one's mirrored block by block
// (band_mirror; the matrix is symmetric: its columns are the first one's rows), so both run the same code; the copies
// are padded with zero / identity columns where a chain would otherwise need a mask (identity pivots in front of the
// mirrored chain make its length a multiple of W).  The first chain takes the W rows in the middle (three whole
// blocks): it adds the second's Schur complement (through LDS, one barrier) and goes on; back substitution from the
// middle outwards likewise.  g and the bands may be assembled by further workgroups of the same launch (penta_pipe.h
// PipeAsm), as in penta_pipe_kernel.
//
// The multiplier rows of a KKT system (kkt.h) are ordinary pivots here - negative ones; they are not tested
// (ldl_pivot_bad), kkt_extract_kernel judges them from Dst.
#pragma once

#include <type_traits>

#include "penta_pipe.h"

namespace idto_dev {

struct BandArgs {
  int n, k;                              // block rows of the system, block size: M = n k unknowns, half width 3 k - 1
  const double *HA, *HB, *HC, *b;        // bands and right-hand side from the system's first block row on
  double rhs_sign;
  double* x;                             // out
  double* Dst;                           // out: 1 / d of every pivot ([row][k], what the other kernels leave there)
  unsigned* status; unsigned fact_id;    // factorisation status, as penta_ldl_kernel's
  unsigned epoch;
  size_t pstride;
  int npos;                              // > 0: pivots [npos, k) of a block row are multiplier rows (not tested)
  double* ts;                            // debug stamps or nullptr
};

// The split and the LDS carve-up (doubles).  First chain: pivots 0 .. m - 1 (m a multiple of W), then the W middle rows
// m .. lim - 1 (three whole blocks; w would do); mirrored chain: the nb rows behind them in the order band_mirror
// gives them, `pad` identity pivots in front.  Columns of a copy: FRONT zero columns (the back substitution's reads
// above row 0 and its blocks of four steps run into them), the chain's own, 2 W + 1 padding columns (identity behind the first chain's, zero behind
// the mirrored chain's: those only collect its Schur complement).
struct BandLds { int m, lim, nb, pad, tcols, bcols, T, Bm, Dt, Db, D0, J, end; };
constexpr int BAND_FRONT = 32;
__host__ __device__ inline BandLds band_layout(int M, int W) {
  BandLds L;
  L.m = ((M - W) / 2 + W / 2) / W * W;
  L.lim = L.m + W;
  L.nb = M - W - L.m;
  L.pad = (W - L.nb % W) % W;
  L.tcols = BAND_FRONT + L.lim + 2 * W + 1;
  L.bcols = BAND_FRONT + L.pad + L.nb + 2 * W + 1;
  int o = 0;
  L.T = o; o += L.tcols * 16;
  L.Bm = o; o += L.bcols * 16;
  L.Dt = o; o += L.tcols;
  L.Db = o; o += L.bcols;
  L.D0 = o; o += M + (M & 1);   // the diagonal entries as assembled (pivot test)
  L.J = o; o += W * 16;   // the mirrored chain's window at the join, [slot][lane]
  L.end = o;
  return L;
}

template <int I, int N, class Fn>
__device__ __forceinline__ void band_static_for(Fn&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); band_static_for<I + 1, N>(f); }
}
// lane l of a row of 16 <- lane l + C of the row, 0 past its end
template <int C>
__device__ __forceinline__ double band_shl(double src) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(src), 0x100 | C, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(src), 0x100 | C, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

// One chain on its copy `arr` (cell (q, d) at arr[16 q + d], q = 0 the chain's first pivot), 1 / d to dinv[q].
template <int W>
struct BandChain {
  static constexpr int w = W - 1;
  double R[W];                  // the window

  __device__ __forceinline__ void init(const double* arr) {
    const int l16 = threadIdx.x & 15;
    band_static_for<0, W>([&](auto Sc) {
      constexpr int S = decltype(Sc)::value;
      R[S] = arr[S * 16 + l16];
    });
  }

  // the groups of W pivots [g0, g1)
  __device__ __forceinline__ void forward(double* arr, double* dinv, int g0, int g1) {
    const int l16 = threadIdx.x & 15;
    const bool is15 = l16 == 15;
    // (column j + W is first touched by pivot j + 1, right after the slot is free: it is fetched a pivot earlier
    // into `nxt` - an LDS round trip is longer than what is left of a pivot)
    double nxt = arr[(g0 * W + W) * 16 + l16];
    for (int g = g0; g < g1; ++g) {
      double* pg = arr + g * W * 16 + l16;
      double* dg = dinv + g * W;
      band_static_for<0, W>([&](auto Sc) {
        constexpr int S = decltype(Sc)::value;
        const double colv = R[S];
        const double dj = rdlane(colv, 0);
        const double inv = fast_rcp(dj);
        const double lp = colv * inv;   // lane d: l_{j + d, j}; lane 15: y_j / d_j
        pg[S * 16] = lp;                // the factor's column (cell 0 is not read again)
        dg[S] = inv;
        // (the pivots are tested afterwards, by all threads: 1 / d is in dinv, the entry it started from in d0)
        const double l0 = is15 ? 0.0 : lp, ly = is15 ? lp : 0.0;
        band_static_for<1, W>([&](auto Cc) {
          constexpr int c = decltype(Cc)::value, SC = (S + c) % W;
          const double cc = rdlane(colv, c);
          const double lsh = band_shl<c>(l0);   // (lane 15 - c gets lane 15's: the zero of l0)
          R[SC] = __builtin_fma(-ly, cc, __builtin_fma(-lsh, cc, R[SC]));
        });
        R[S] = nxt;
        nxt = pg[(S + W + 1) * 16];
      });
    }
  }

  // rows j1 - 1 .. j0 in blocks of four steps (the last block may run up to three rows past j0: the chain's next rows,
  // or padding): x into cell 15.  P: lane d = what the rows above j have added to row j - d so far.  Returns the row
  // the next call starts from (its j1).
  static __device__ __forceinline__ int backward(double* arr, int j1, int j0, double& P) {
    constexpr int U = 4;
    const int l16 = threadIdx.x & 15;
    // lane d reads cell (j - d, d) = 16 j - 15 d; lane 15 cell (j, 15)
    const double* pr = arr + (l16 == 15 ? 15 : -15 * l16);
    // (lane 0's cell (j, 0) needs no care: what it adds to p_0 is shifted out; lane 15's is y_j / d_j, no factor)
    const double keep = l16 == 15 ? 0.0 : 1.0;
    double Ln[U], zn[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const double raw = pr[(j1 - 1 - u) * 16];
      zn[u] = rdlane(raw, 15);
      Ln[u] = raw * keep;
    }
    int jt = j1 - 1;
    for (; jt >= j0; jt -= U) {
      double Lc[U], zc[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { Lc[u] = Ln[u]; zc[u] = zn[u]; }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const double raw = pr[(jt - U - u) * 16];
        zn[u] = rdlane(raw, 15);
        Ln[u] = raw * keep;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const double xj = zc[u] + rdlane(P, 0);
        arr[(jt - u) * 16 + 15] = xj;
        P = band_shl<1>(__builtin_fma(-Lc[u], xj, P));
      }
    }
    return jt + 1;
  }
};

// where entry (p, q), p >= q, of the band matrix is in the blocks (column-major K x K; C_t both triangles); in: it exists
// (offset inside the band array `band`: 0 C, 1 B, 2 A)
__host__ __device__ inline int band_entry_off(int KB, int p, int q, bool& in, int& band) {
  const int t = p / KB, r = p - t * KB, s = q / KB, c = q - s * KB, bd = t - s;
  in = bd <= 2;
  band = bd;
  return t * (KB * KB) + c * KB + r;
}
template <int KB>
__device__ __forceinline__ const double* band_entry(const BandArgs& A, int p, int q, bool& in) {
  int bd;
  const int off = band_entry_off(KB, p, q, in, bd);
  const double* blk = bd == 0 ? A.HC : bd == 1 ? A.HB : A.HA;
  return blk + off;
}

// The mirrored chain's order: the BLOCKS from the last one backwards, the rows inside a block as they are - a KKT
// system's multiplier rows still come behind their block's variables (the other way round they would be zero pivots;
// kkt.h).  The map is its own inverse; the half width stays 3 K - 1.
__host__ __device__ inline int band_mirror_rt(int KB, int i, int M) {
  const int t = i / KB;
  return (M / KB - 1 - t) * KB + (i - t * KB);
}
template <int KB>
__device__ __forceinline__ int band_mirror(int i, int M) { return band_mirror_rt(KB, i, M); }

// The staging pass as a TABLE (gn_small.h: the bands sit in the workgroup's own LDS, the index arithmetic below - seven
// integer divisions an item - was 1.3 / 2.5 us of acrobot's / the spinner's step): for item e of penta_band_body's staging
// loop {source, is the right-hand side, destination, second destination (the diagonal entry by row) or -1}; source: the
// element's offset in [A | B | C | b] laid out as `offA, offB, offC, offb` say, or -1 for a zero.  Same decisions, same order.
struct BandStageItem { int src, rhs, dst, dst0; };
inline int band_stage_table(int n, int KB, int offA, int offB, int offC, int offb, BandStageItem* out) {
  const int W = 3 * KB, M = n * KB;
  const BandLds L = band_layout(M, W);
  const int nitem = (L.lim + L.nb) * (W + 1);
  if (!out) return nitem;
  for (int e = 0; e < nitem; ++e) {
    const int qc = e / (W + 1), d = e - qc * (W + 1);
    const bool second = qc >= L.lim;
    const int i = second ? qc - L.lim : qc;
    const bool rhs = d == W;
    const int dd = rhs ? 0 : d;
    const bool ent = !rhs && i + dd < (second ? M : L.lim);
    const int pa = second ? band_mirror_rt(KB, ent ? i + dd : i, M) : i + dd, pb = second ? band_mirror_rt(KB, i, M) : i;
    const int hi = pa > pb ? pa : pb, lo = pa > pb ? pb : pa;
    bool blk_in = false;
    int bd = 0;
    const int off = band_entry_off(KB, ent ? hi : 0, ent ? lo : 0, blk_in, bd);
    BandStageItem it;
    it.rhs = rhs ? 1 : 0;
    it.src = rhs ? offb + pb : ((ent && blk_in) ? (bd == 0 ? offC : bd == 1 ? offB : offA) + off : -1);
    it.dst = (second ? L.Bm + (BAND_FRONT + L.pad + i) * 16 : L.T + (BAND_FRONT + i) * 16) + (rhs ? 15 : d);
    it.dst0 = d == 0 ? L.D0 + pb : -1;
    out[e] = it;
  }
  return nitem;
}

// the solver's workgroup (problem = blockIdx.y): wavefronts 0 / 1 = the chains, every wavefront stages; any block size of
// at least two wavefronts (penta_band_kernel: 256 threads; gn_small.h calls it at the end of its 512-thread workgroup,
// with F.on = 0: g and the bands are in memory)
// the copies' padding: zeros, a one on the diagonal of the identity columns (the first chain's from lim on, the mirrored
// chain's pad); the cells of real columns the staging loads do not write (16 - W - 2 of them per column) are zero too
__device__ __forceinline__ void band_pad(double* lds, const BandLds& L, int tid, int nt) {
  for (int e = tid; e < (L.tcols + L.bcols) * 16; e += nt) {
    const bool second = e >= L.tcols * 16;
    const int ee = second ? e - L.tcols * 16 : e, col = (ee >> 4) - BAND_FRONT, d = ee & 15;
    const bool ident = second ? (col >= 0 && col < L.pad) : col >= L.lim;
    lds[L.T + e] = (ident && d == 0) ? 1.0 : 0.0;
  }
}

// PADDED: the caller has run band_pad behind a barrier of its own (gn_small.h: with its first loads); STAGED: and filled
// the copies (band_stage_table), likewise
template <int W, bool PADDED = false, bool STAGED = false>
__device__ __forceinline__ void penta_band_body(BandArgs A, const PipeAsm F) {
  extern __shared__ double lds[];
  constexpr int w = W - 1, KB = W / 3;
  const size_t o = (size_t)blockIdx.y * A.pstride;
  A.HA = at_problem(A.HA, o); A.HB = at_problem(A.HB, o); A.HC = at_problem(A.HC, o); A.b = at_problem(A.b, o);
  A.x = at_problem(A.x, o); A.Dst = at_problem(A.Dst, o);
  A.status += 2 * blockIdx.y;
  const int tid = threadIdx.x, nt = blockDim.x, wave = tid >> 6, lane = tid & 63, l16 = tid & 15;
  const int M = A.n * KB;
  const BandLds L = band_layout(M, W);
  const int m = L.m, lim = L.lim, nb = L.nb, pad = L.pad;
  if (A.ts && tid == 0) A.ts[0] = (double)wall_clock64();
  // the copies' padding while the assembly (if it is this launch's) is still under way
  if (!PADDED) {
    band_pad(lds, L, tid, nt);
    __syncthreads();
  }
  if (pipe_asm_on(F, o)) {   // g and the bands come from this very launch: all of them, then plain loads
    const unsigned* ready = at_problem(F.ready, o);
    const SpinCtl sc{A.status + 2 * (gridDim.y - blockIdx.y), A.fact_id};
    for (int i = tid; i < 4 * F.rows; i += nt)
      spin_wait([&] { return __hip_atomic_load(ready + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == A.epoch; }, sc);
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  if (A.ts && tid == 0) A.ts[1] = (double)wall_clock64();
  // ---- the two copies: loads of the entries that exist, eight per thread in flight (addresses clamped, results
  // selected: no load behind a branch).  Item: column (the first chain's lim, then the mirrored chain's nb), cell of
  // the column (0 .. w: the entry, W: the right-hand side -> cell 15)
  const int nitem = STAGED ? 0 : (lim + nb) * (W + 1);
  constexpr int NB = W <= 6 ? 4 : (W <= 9 ? 8 : 16);   // (one batch covers horizons of ~45 steps)
  for (int e0 = tid; e0 < nitem; e0 += NB * nt) {
    double v[NB];
    int dst[NB], dst0[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int e = e0 + u * nt, ec = e < nitem ? e : 0;
      const int qc = ec / (W + 1), d = ec - qc * (W + 1);
      const bool second = qc >= lim;
      const int i = second ? qc - lim : qc;   // the chain's index
      // first copy: entry (i + d, i), rows < lim only.  Mirrored: the rows band_mirror(i + d), band_mirror(i)
      const bool rhs = d == W;
      const int dd = rhs ? 0 : d;
      const bool ent = !rhs && i + dd < (second ? M : lim);
      const int pa = second ? band_mirror<KB>(ent ? i + dd : i, M) : i + dd, pb = second ? band_mirror<KB>(i, M) : i;
      const int hi = pa > pb ? pa : pb, lo = pa > pb ? pb : pa;
      bool blk_in = false;
      const double* src = band_entry<KB>(A, ent ? hi : 0, ent ? lo : 0, blk_in);
      const double* ptr = rhs ? A.b + pb : src;
      const double val = *ptr * (rhs ? A.rhs_sign : 1.0);
      v[u] = (rhs || (ent && blk_in)) ? val : 0.0;
      dst[u] = e < nitem ? (second ? L.Bm + (BAND_FRONT + pad + i) * 16 : L.T + (BAND_FRONT + i) * 16) + (rhs ? 15 : d) : -1;
      dst0[u] = (e < nitem && d == 0) ? L.D0 + pb : -1;   // the diagonal entry once more, by row (pivot test)
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) asm volatile("" : "+v"(v[u]));
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      if (dst[u] >= 0) lds[dst[u]] = v[u];
      if (dst0[u] >= 0) lds[dst0[u]] = v[u];
    }
  }
  if (!STAGED) __syncthreads();
  if (A.ts && tid == 0) A.ts[2] = (double)wall_clock64();
  double* Tc = lds + L.T + BAND_FRONT * 16;    // column 0 of the first chain's copy
  double* Bc = lds + L.Bm + BAND_FRONT * 16;   // index 0 (the first identity pivot) of the mirrored chain's
  if (wave == 0) {
    BandChain<W> ch;
    ch.init(Tc);
    ch.forward(Tc, lds + L.Dt + BAND_FRONT, 0, m / W);
    __syncthreads();   // the mirrored chain's window is in J
    {
      // row m + x of the middle is the mirrored chain's index nb + u(x), u(x) = (2 - x / K) K + x mod K (band_mirror),
      // column nb + u in its slot u (nb + pad is a multiple of W), the entry (nb + u1, nb + u2), u1 >= u2, in lane u1 - u2
      auto um = [&](int x) { const int blk = x / KB; return (2 - blk) * KB + (x - blk * KB); };
      band_static_for<0, W>([&](auto Sc) {
        constexpr int S = decltype(Sc)::value;
        // slot S holds column m + S: lane d its entry (m + S + d, m + S), lane 15 the right-hand side of row m + S
        const int d = l16 == 15 ? 0 : l16;
        const bool in = S + d < W && d <= w;
        const int ua = um(in ? S + d : S), ub = um(S), u1 = ua > ub ? ua : ub, u2 = ua > ub ? ub : ua;
        const double add = lds[L.J + u2 * 16 + (l16 == 15 ? 15 : u1 - u2)];
        ch.R[S] += in ? add : 0.0;
      });
    }
    ch.forward(Tc, lds + L.Dt + BAND_FRONT, m / W, m / W + 1);
    if (A.ts && tid == 0) A.ts[3] = (double)wall_clock64();
    double P = 0.0;
    const int jn = BandChain<W>::backward(Tc, lim, m, P);   // the middle rows (and, to fill its last block, up to three more)
    __syncthreads();                                          // ... their x is in the copy's cells 15
    BandChain<W>::backward(Tc, jn, 0, P);
  } else if (wave == 1) {
    BandChain<W> ch;
    ch.init(Bc);
    ch.forward(Bc, lds + L.Db + BAND_FRONT, 0, (nb + pad) / W);
    band_static_for<0, W>([&](auto Sc) {
      constexpr int S = decltype(Sc)::value;
      if (lane < 16) lds[L.J + S * 16 + lane] = ch.R[S];
    });
    __syncthreads();
    __syncthreads();
    // the middle rows (index nb + pad + c = row band_mirror(nb + c) of the first chain's copy, x in its cell 15) push into
    // the rows above them through this chain's factors; their own columns here hold nothing
    double P = 0.0;
    for (int c = W - 1; c >= 0; --c) {
      const int j = nb + pad + c;
      const double xj = Tc[band_mirror<KB>(nb + c, M) * 16 + 15];
      const double Lr = (l16 >= 1 && l16 <= w) ? Bc[j * 16 - 15 * l16] : 0.0;
      P = band_shl<1>(__builtin_fma(-Lr, xj, P));
    }
    BandChain<W>::backward(Bc, nb + pad, pad, P);
  } else {
    __syncthreads();
    __syncthreads();
  }
  __syncthreads();
  if (A.ts && tid == 0) A.ts[4] = (double)wall_clock64();
  // results out; and the pivot test (penta_ldl.h ldl_pivot_bad): positive, finite, not cancelled to nothing against the
  // entry it started from (NaN fails both comparisons); a KKT system's multiplier rows are judged elsewhere
  bool bad = false;
  for (int j = tid; j < M; j += nt) {
    const bool first = j < lim;
    const int jm = pad + band_mirror<KB>(j, M);
    A.x[j] = first ? Tc[j * 16 + 15] : Bc[jm * 16 + 15];
    const double inv = first ? lds[L.Dt + BAND_FRONT + j] : lds[L.Db + BAND_FRONT + jm];
    A.Dst[j] = inv;
    const bool multiplier = A.npos > 0 && j % KB >= A.npos;
    bad = bad || (!multiplier && !(inv > 0.0 && inv * lds[L.D0 + j] < 4503599627370496.0));
  }
  if (__builtin_amdgcn_ballot_w64(bad) != 0ull && lane == 0) {
    __hip_atomic_store(A.status, A.fact_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_fetch_add(A.status + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// grid: (1 + assembly workgroups, problems); 256 threads
template <int W>
__global__ void __launch_bounds__(256) penta_band_kernel(BandArgs A, PipeAsm F) {
  if (blockIdx.x >= 1) { pipe_assemble(A.ts, A.epoch, A.pstride, F, 1); return; }
  penta_band_body<W>(A, F);
}

}  // namespace idto_dev
