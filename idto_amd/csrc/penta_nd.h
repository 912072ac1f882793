// penta_nd.h — nested-dissection form of the block LDL^T factor + solve (penta_ldl.h): the
// dependent chain of the two-workgroup kernel (n/2 block rows forward, n/2 backward) is cut again.
//
//   rows:  0 ........ j1-1 | j1 j1+1 | j1+2 ........ s-1 | s s+1 | s+2 ........ j2-1 | j2 j2+1 | j2+2 ........ n-1
//          chain P0 (down)    join 1    chain J1 (up)      separator  chain J2 (down)    join 2    chain P3 (up)
//
// Four workgroups run the row recursion of penta_ldl_body concurrently: P0 / P3 are "producers"
// (plain ends of the matrix, they finish with two product-only pseudo-rows = their Schur
// contributions to the join rows), J1 / J2 are "joiners" (they start next to the separator, add the
// producer's contributions and eliminate the two join rows) - exactly the two roles of the
// two-workgroup kernel, twice.  What is new is the coupling of J1 / J2 to the separator rows s, s+1:
// eliminating a chain that is coupled to rows outside of it fills the coupling blocks
//   F_il = [ H(row il, sep row nearest) | H(row il, sep row farthest) ]            (K x 2K, nonzero for il = 0, 1)
//   Ft_il = L_il^-1 ( F_il - (D^-1 Ht_{il-1})^T Ft_{il-1} - (D^-1 Et_{il-2})^T Ft_{il-2} )
// which behave like 2K more right-hand sides.  They are NOT carried by the chain workgroups (their
// row time would grow by half): a "spike" workgroup per joiner trails it by one block row, reading
// each row's factors from HBM when the joiner's I/O wavefronts have released them (per-row counter),
// keeps every Ft_il in LDS, accumulates  Q = sum_il Ft_il^T Dn_il Ft_il  and  qy = sum_il Ft_il^T Dn_il rt_il
// on the matrix cores, and hands both to the separator workgroup, which eliminates
//   [ C_s - Q..   B_{s+1}^T - Q.. ] [x_s    ]   [ r_s     - qy.. ]
//   [ .           C_{s+1} - Q..   ] [x_{s+1}] = [ r_{s+1} - qy.. ]
// solves it and publishes x_s, x_{s+1}.  The spike workgroups then form F_il^T-side corrections
// Ft_il [x_sep] for all their rows at once; the joiners subtract them from rt and run the usual back
// substitution (join rows first, handing x_join to the producers).
// Dependent chain for n = 40, K = 19: 9 + 2 rows, one trailing spike row, the separator (2 rows) and
// 2 + 9 rows back, instead of 20 + 2 and 22.
#pragma once
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "penta_nd.h / penta_pipe.h / penta_ldl.h order their row hand-overs by write-through stores + s_waitcnt vmcnt(0) (see release_row): gfx942 / gfx950 only"
#endif

#include <type_traits>
#include "penta_ldl.h"

namespace idto_dev {

enum { ND_MAXROWS = 32 };  // local rows of a joiner chain (incl. its two join rows)

struct NdArgs {
  int n, k;
  const double* HA; const double* HB; const double* HC; const double* b;
  double rhs_sign;
  double* x;
  double* Ust; double* Hst; double* Est; double* Dst;
  int s, j1, j2;                 // separator rows s, s+1; join rows j1, j1+1 and j2, j2+1
  double* xch; int xch_pair;     // two producer/joiner exchange buffers (layout of the two-workgroup kernel)
  unsigned* flags;               // [0..1] pair P0/J1, [2..3] pair P3/J2, [4] separator solved
  unsigned long long* rowcnt;    // [4][ND_MAXROWS]: released wavefronts per local row of J1, J2 (chains) and of their spike workgroups
  unsigned long long rowtarget;  // value a joiner row's counter reaches in this launch ...
  unsigned long long rowunit;    // ... = launches x rowunit (I/O wavefronts of a chain workgroup)
  double* ndbuf;                 // see nd_layout
  unsigned epoch;
  unsigned* status; unsigned fact_id;
  size_t pstride;
  double* ts;                    // option "solver_debug": [7 roles][64] wall-clock stamps (100 MHz)
  SpinCtl spin;                  // where a wait between workgroups that ran out reports it (penta_ldl.h spin_wait)
  int debug_skip_role;           // test aid: this role returns at once (its partners' waits must run out), -1: none
  const unsigned* asm_ready;     // penta_pipe.h PipeAsm: the launch assembles g and the bands itself ([rows][4] epoch words), else nullptr
  int asm_first;
  int wt_rows;                   // 1 (penta_pipe_kernel): the chains publish a row's spike block, 1 / d and rt with write-through stores, the separator reads past the L2 (no acquire);
                                 // 0 (penta_nd_kernel): write-through stores, the readers acquire and read with plain loads; 2: plain stores and a releasing fence (measurement aid IDTO_ND_WT)
  int debug_pipe_tail;           // measurement aid: penta_pipe_kernel takes the row-by-row back substitution
  int npos;                      // > 0: a KKT system (kkt.h) - the pivots [npos, k) of every block row are not tested (ldl_pivot_bad)
  int lds_rows;                  // the chains' carve-up holds this many local rows (penta_ldl_layout `rows`; 0: all n)
  // back substitution in recursion form (penta_pipe.h chain_recursion_tail; K > 20): the launch has lds_doubles of LDS
  // and wst holds [2][ND_MAXROWS] rows of nd_layout(K).frow doubles for the joiners' W_il
  int rec_tail, lds_doubles;   // (rec_tail: how many of a producer's wavefronts form its joiner's W rows, pipe_recursion_tail_fits)
  double* wst;
};
__device__ __forceinline__ void nd_ts(const NdArgs& A, int role, int slot) {
  if (A.ts && threadIdx.x == 0) A.ts[role * 64 + slot] = (double)wall_clock64();
}

struct NdBuf { int rtpub, fst, frow, xsep, ll, joinll, joinll_pair, end; };  // offsets in doubles; rtpub / fst are [2][...]
__host__ __device__ inline NdBuf nd_layout(int K) {
  NdBuf L;
  const int ks = ldl_ks(K), ct2 = (2 * K + 1 + 15) / 16;   // columns [Ft | rt], in tiles of 16
  int o = 0;
  L.rtpub = o; o += 2 * ND_MAXROWS * K;
  L.frow = 16 * ct2 * ks;                // doubles per published row (whole 16-column tiles: the separator's MFMA loads)
  L.fst = o; o += 2 * ND_MAXROWS * L.frow + 16 * ks;
  L.xsep = o; o += 2 * K;
  o += o & 1;
  L.ll = o; o += 3 * 4 * K;              // flagged copies (ll_store): x_sep, and the two join rows of each producer / joiner pair
  L.joinll = o; L.joinll_pair = 2 * 2 * K * 64;            // [pseudo row][r][column: 64 lanes] of 16-byte slots   // penta_pipe.h: a producer's Schur-complement contributions to the join rows, flagged
  o += 2 * L.joinll_pair;
  L.end = o;
  return L;
}

// tile t of the lower triangle of a tile grid, row by row: (0,0) (1,0) (1,1) (2,0) (2,1) (2,2) ...
__host__ __device__ constexpr int nd_tile_row(int t) { int tr = 0; while (t > tr) { t -= tr + 1; ++tr; } return tr; }
__host__ __device__ constexpr int nd_tile_col(int t) { int tr = 0; while (t > tr) { t -= tr + 1; ++tr; } return t; }
// f(integral_constant<int, I>) for I = I0 .. N-1: the tile loops below index register arrays with nd_tile_row / _col of
// the loop variable, and left as functions of a run-time t the compiler EVALUATED THEM IN LOOPS and went through
// scratch for the arrays (K = 29: 7 us for the 80 matrix-core instructions of a separator row)
template <int I, int N, class F>
__device__ __forceinline__ void nd_static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); nd_static_for<I + 1, N>(f); }
}

__device__ __forceinline__ void nd_wait(const unsigned* f, unsigned epoch, const SpinCtl sc) {
  if (threadIdx.x == 0)
    spin_wait([&] { return __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch; }, sc);
  __syncthreads();
  (void)__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
}
// A double handed to another workgroup in ONE memory round trip: each 32-bit half travels in an 8-byte word with the
// launch's epoch next to it (8-byte stores are single transactions), so the reader polls the data itself - no flag to
// wait for first, no cache invalidation before the data may be read.  (The usual flag + acquire + load costs three.)
__device__ __forceinline__ void ll_store(double* slot2, double v, unsigned epoch) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v), e = (unsigned long long)epoch << 32;
  // (agent-scope stores: a plain store may sit dirty in this XCD's L2, where the reader's XCD does not look)
  unsigned long long* q = reinterpret_cast<unsigned long long*>(slot2);
  __hip_atomic_store(q, (b & 0xffffffffull) | e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(q + 1, (b >> 32) | e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool ll_try(const double* slot2, unsigned epoch, double& v) {
  const unsigned long long* q = reinterpret_cast<const unsigned long long*>(slot2);
  const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v = __longlong_as_double((long long)((a & 0xffffffffull) | (b << 32)));
  return (unsigned)(a >> 32) == epoch && (unsigned)(b >> 32) == epoch;
}
__device__ __forceinline__ double ll_load(const double* slot2, unsigned epoch, const SpinCtl& sc) {
  double v = 0.0;
  spin_wait([&] { return ll_try(slot2, epoch, v); }, sc);
  return v;
}

__device__ __forceinline__ void nd_post(unsigned* f, unsigned epoch) {
  __syncthreads();   // every wavefront's global stores precede thread 0's release
  if (threadIdx.x == 0) __hip_atomic_store(f, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- spike workgroup of joiner `w` (0: J1, mirrored, starts at row s-1; 1: J2, starts at row s+2)
// Per row il of the joiner (its factors and rt released through the row counter):
//   phase 1 (all wavefronts)  F_il = Finit - (D^-1 Ht_{il-1})^T Ft_{il-1} - GE          (matrix cores)
//   phase 2  wavefront 0      Ft_il = L_il^-1 F_il                                        (lane = column)
//            wavefronts 1..3  GE = (D^-1 Et_{il-1})^T Ft_{il-1} for row il+1, and [Ft_{il-1} | rt_{il-1}] to HBM
//                             for the separator (Q) and the joiner (correction), released per row
// Only the dependent pair (phase 1, substitution) is on the row's critical path: ~3.2 us, the
// joiner needs ~3.6 us per row, so this workgroup stays one row behind it.
template <int K, bool PADDED>
__device__ __forceinline__ void nd_spike(const NdArgs& A, const int w) {
  extern __shared__ double lds[];
  constexpr int ks = ldl_ks(K), SK = (K + 3) / 4, TT = (K + 15) / 16, NF = 2 * K, CT = (NF + 15) / 16, KP = 4 * SK;
  using d4 = __attribute__((ext_vector_type(4))) double;
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6;
  const int k = A.k, kk = k * k;
  const bool mirror = (w == 0);
  const int base = mirror ? A.s - 1 : A.s + 2;
  const int nloc = mirror ? A.s - A.j1 : A.j2 + 2 - (A.s + 2);   // chain rows + the two join rows
  auto orig = [&](int il) { return mirror ? base - il : base + il; };
  const NdBuf B = nd_layout(K);
  const unsigned long long* rowcnt = A.rowcnt + w * ND_MAXROWS;
  unsigned long long* frow = A.rowcnt + (2 + w) * ND_MAXROWS;
  const double* rtpub = A.ndbuf + B.rtpub + (size_t)w * ND_MAXROWS * K;
  double* Fst = A.ndbuf + B.fst + (size_t)w * ND_MAXROWS * B.frow;
  // LDS (doubles)
  constexpr int FB = NF * ks;             // one Ft block: NF columns of K rows, stride ks (pad rows zero)
  double* Fr = lds;                       // [2][FB] ring: Ft_il, Ft_{il-1}
  double* Ul = Fr + 2 * FB;               // [KP][ks] row-major D^-1 U_il (strict upper), zero pad rows
  double* Hl = Ul + KP * ks;              // D^-1 Ht_{il-1}
  double* El = Hl + KP * ks;              // D^-1 Et_{il-1}
  double* Fi = El + KP * ks;              // [3][K*K] initial coupling blocks, column-major
  double* rt = Fi + 3 * K * K + (K & 1);  // [2][ks] rt_il, rt_{il-1}
  double* GE = rt + 2 * ks;               // [FB] (D^-1 Et_{il-1})^T Ft_{il-1}: the second product of row il+1, a row early
  for (int idx = tid; idx < 2 * FB + 3 * KP * ks; idx += nt) lds[idx] = 0.0;
  for (int idx = tid; idx < FB + 2 * ks; idx += nt) rt[idx] = 0.0;   // (rt and GE are adjacent)
  // initial coupling blocks (rows il = 0, 1 of the chain to the separator rows -1, -2 in its own orientation):
  //   Fi[0] = coupling(row 0, row -1), Fi[1] = coupling(row 0, row -2), Fi[2] = coupling(row 1, row -1)
  for (int idx = tid; idx < 3 * K * K; idx += nt) {
    const int blk = idx / (K * K), e = idx - blk * K * K, c = e / K, r = e - c * K;
    double val = 0.0;
    if (r < k && c < k) {
      if (!mirror) {   // rows base, base+1; separator rows base-1 (near), base-2 (far): B_base, A_base, A_{base+1} as stored
        const double* src = (blk == 0) ? A.HB + (size_t)base * kk : (blk == 1) ? A.HA + (size_t)base * kk : A.HA + (size_t)(base + 1) * kk;
        val = src[c * k + r];
      } else {         // rows base, base-1; separator rows base+1, base+2: B_{base+1}^T, A_{base+2}^T, A_{base+1}^T
        const double* src = (blk == 0) ? A.HB + (size_t)(base + 1) * kk : (blk == 1) ? A.HA + (size_t)(base + 2) * kk : A.HA + (size_t)(base + 1) * kk;
        val = src[r * k + c];
      }
    }
    Fi[idx] = val;
  }
  __syncthreads();
  nd_ts(A, 4 + w, 0);
  const int fl = lane & 15, fk = lane >> 4;

  // Factors of row il+1 are fetched into registers while row il is processed (the loads of a row
  // cost ~2.5 us exposed, which alone would make this workgroup slower than the chain it follows).
  constexpr int BLK = K * ks, PM = (3 * BLK + 255) / 256;
  double pre[PM], prt = 0.0;
  auto row_ready = [&](int il) {   // uniform: every thread loads the same counter
    return __hip_atomic_load(rowcnt + il, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= A.rowtarget;
  };
  auto issue = [&](int il) {       // (after an acquire on row il's counter)
    const double* Ug = A.Ust + (size_t)orig(il) * BLK;
    const double* Hg = A.Hst + (size_t)orig(il >= 1 ? il - 1 : 0) * BLK;
    const double* Eg = A.Est + (size_t)orig(il >= 1 ? il - 1 : 0) * BLK;   // (for row il+1's second product)
#pragma unroll
    for (int sl = 0; sl < PM; ++sl) {
      const int idx = tid + sl * 256, blk = idx / BLK, e = idx - blk * BLK;
      double v = 0.0;
      if (blk == 0) v = Ug[e];
      else if (blk == 1) { if (il >= 1) v = Hg[e]; }
      else if (blk == 2) { if (il >= 1) v = Eg[e]; }
      pre[sl] = v;
    }
    if (tid < ks) prt = (tid < K) ? rtpub[(size_t)il * K + tid] : 0.0;
  };
  auto commit = [&](int il) {      // registers -> LDS (Ul, Hl, El are KP * ks apart)
#pragma unroll
    for (int sl = 0; sl < PM; ++sl) {
      const int idx = tid + sl * 256, blk = idx / BLK, e = idx - blk * BLK;
      if (blk < 3) Ul[blk * KP * ks + e] = pre[sl];
    }
    if (tid < ks) rt[(il & 1) * ks + tid] = prt;
  };
  // (per wavefront, no workgroup barrier: the wavefronts decide independently whether they could
  // prefetch a row, so nothing here may assume that the others take the same path)
  auto wait_row = [&](int il) {
    spin_wait([&] { return row_ready(il); }, A.spin);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  };
  // [Ft_il | rt_il] -> HBM (row stride B.frow, column stride ks), by the threads [t0, t0 + tn)
  auto publish = [&](int il, int t0, int tn) {
    const double* F = Fr + (il & 1) * FB;
    double* dst = Fst + (size_t)il * B.frow;
    for (int idx = tid - t0; idx < FB + ks; idx += tn) {
      const double v = (idx < FB) ? F[idx] : rt[(il & 1) * ks + idx - FB];
      // (write-through: nothing of the row stays dirty in this XCD's L2, and the release below needs no write-back of
      // that L2 - three of those per row from each of 64 spike workgroups cost a batch of 32 problems 12% of its
      // throughput.  The readers invalidate and read with plain loads: reading past the L2 word by word, as the
      // pipelined kernel's separator does, took the separator 2 us longer per row at K = 23.)
      if (A.wt_rows != 2) __hip_atomic_store(dst + idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else dst[idx] = v;
    }
  };
  // HARDWARE ASSUMPTION (gfx942 / gfx950 only, enforced by the #error at the top of this file): an agent-scope relaxed
  // atomic store is a `global_store ... sc1` that writes through this XCD's L2, and `s_waitcnt vmcnt(0)` returns once
  // every such store has been acknowledged at the agent's coherence point.  That is what LLVM's own release sequence
  // for these targets ends with (AMDGPUUsage, memory model gfx942: `buffer_wbl2 sc1; s_waitcnt vmcnt(0)`) minus the
  // write-back of the whole L2 - which has nothing of this row to write back, every store of the row being
  // write-through.  The relaxed increment that follows is therefore ordered behind the row at agent scope by the
  // hardware, not by the HIP memory model; readers poll it, execute an acquiring fence (L2 invalidate) and load.
  // IDTO_ND_WT=2 brings the formal release back (plain stores + a releasing fence): tools/nd_stress.py and
  // tools/stress_solver.py run both and compare bits.
  auto release_row = [&](int il) {   // by each of the three wavefronts that stored a part of the row
    if (A.wt_rows != 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (lane == 0) __hip_atomic_fetch_add(frow + il, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  wait_row(0);
  issue(0);
  // readiness of the next row: a round trip to L2 (~1 us, longer than the products of phase 1), asked a row ahead and
  // consumed after the products.  (What it says is a row old: a workgroup that has caught up with its joiner finds
  // the row in wait_row below instead.)
  unsigned long long next_cnt = (1 < nloc) ? __hip_atomic_load(rowcnt + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
  for (int il = 0; il < nloc; ++il) {
    nd_ts(A, 4 + w, 8 + 2 * il);
    commit(il);
    __syncthreads();
    if (il == 8) nd_ts(A, 4 + w, 44);
    // (second chance for a row the look-ahead missed: asked now, looked at after the products only in that case)
    const unsigned long long fresh_cnt =
        (il + 1 < nloc) ? __hip_atomic_load(rowcnt + il + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    // ---- phase 1: F_il = Finit - GE - (D^-1 Ht_{il-1})^T Ft_{il-1}: TT x CT tiles
    {
      const double* F1 = Fr + ((il + 1) & 1) * FB;
      double* Fo = Fr + (il & 1) * FB;
      auto put = [&](int tr, int tc, const d4& acc) __attribute__((always_inline)) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int r = 16 * tr + fk + 4 * rg, c = 16 * tc + fl;
          if (r < K && c < NF) {
            double init = 0.0;
            if (il == 0) init = Fi[(c < K ? 0 : 1) * K * K + (c < K ? c : c - K) * K + r];
            else if (il == 1 && c < K) init = Fi[2 * K * K + c * K + r];
            Fo[c * ks + r] = (init - GE[c * ks + r]) - acc[rg];
          }
        }
      };
      if (TT == 2 && CT <= 4) {
        // blocks of 17 .. 32: a column tile per wavefront, its two row tiles in two accumulators (the six products of
        // a tile are a dependent chain of matrix-core instructions; 6 tiles dealt out over 4 wavefronts were two such
        // chains one after the other on two of them)
        if (wave < CT) {
          const int tc = wave;
          d4 a0 = {0.0, 0.0, 0.0, 0.0}, a1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int sq = 0; sq < SK; ++sq) {
            const int kr = 4 * sq + fk;
            const double f = F1[(16 * tc + fl) * ks + kr];
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Hl[kr * ks + fl], f, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Hl[kr * ks + 16 + fl], f, a1, 0, 0, 0);
          }
          put(0, tc, a0);
          put(1, tc, a1);
        }
      } else {
        for (int t = wave; t < TT * CT; t += nt / 64) {
          const int tr = t / CT, tc = t - tr * CT;
          d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int sq = 0; sq < SK; ++sq) {
            const int kr = 4 * sq + fk;   // summation index: row of Ht and of Ft
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Hl[kr * ks + 16 * tr + fl], F1[(16 * tc + fl) * ks + kr], acc, 0, 0, 0);
          }
          // (F1 is overwritten below only in its own slot's next use; Fo is the other ring slot)
          put(tr, tc, acc);
        }
      }
    }
    bool prefetched = false;
    if (il == 8) nd_ts(A, 4 + w, 49);
    bool next_ready = next_cnt >= A.rowtarget;
    if (!next_ready) next_ready = fresh_cnt >= A.rowtarget;
    if (il + 1 < nloc && next_ready) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      issue(il + 1);
      prefetched = true;
    }
    next_cnt = (il + 2 < nloc) ? __hip_atomic_load(rowcnt + il + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    if (il == 8) nd_ts(A, 4 + w, 45);
    __syncthreads();
    if (il == 8) nd_ts(A, 4 + w, 46);
    // ---- phase 2
    if (wave == 0) {
      // Ft_il = L_il^-1 F_il (L^T = D^-1 U, unit): lane = column, rows in registers, multipliers are LDS broadcasts
      if (lane < NF) {
        double* col = Fr + (il & 1) * FB + lane * ks;
        double xr[K];
#pragma unroll
        for (int r = 0; r < K; ++r) xr[r] = col[r];
        // multipliers: row j of D^-1 U, two per LDS read (ks is even), the next row's read while this one is applied
        // (left alone the compiler reads all K (K - 1) / 2 of them first: scratch)
        constexpr int NP = (K + 1) / 2;
        double2 ub[2][NP];
        auto loadu = [&](int j, double2 (&u)[NP]) __attribute__((always_inline)) {
          const double2* u2 = reinterpret_cast<const double2*>(Ul + j * ks);
#pragma unroll
          for (int r2 = (j + 1) / 2; r2 < NP; ++r2) u[r2] = u2[r2];
        };
        loadu(0, ub[0]);
#pragma unroll
        for (int j = 0; j < K - 1; ++j) {
          if (j + 1 < K - 1) loadu(j + 1, ub[(j + 1) & 1]);
          asm volatile("" ::: "memory");
#pragma unroll
          for (int r2 = (j + 1) / 2; r2 < NP; ++r2) {
            const double2 m = ub[j & 1][r2];
            if (2 * r2 > j) xr[2 * r2] = __builtin_fma(-m.x, xr[j], xr[2 * r2]);
            if (2 * r2 + 1 < K) xr[2 * r2 + 1] = __builtin_fma(-m.y, xr[j], xr[2 * r2 + 1]);
          }
        }
#pragma unroll
        for (int r = 0; r < K; ++r) col[r] = xr[r];
      }
    } else if (il >= 1) {
      // (the stores first, the products while they travel, then the release: the fence waits for the stores' acknowledgements)
      // [Ft_{il-1} | rt_{il-1}] to HBM, released by each of the three wavefronts for its own stores (the separator
      // and the pair's producer wait for all three).  (Up to round 4 the release was one read-modify-write a row
      // later, after a workgroup barrier: the separator saw a row two rows - 7 us at K = 23 - after it was computed.)
      publish(il - 1, 64, 192);
      // GE for row il+1: (D^-1 Et_{il-1})^T Ft_{il-1} (El holds Est of row il-1), tiles over wavefronts 1..3
      const double* F1 = Fr + ((il + 1) & 1) * FB;
      for (int t = wave - 1; t < TT * CT; t += 3) {
        const int tr = t / CT, tc = t - tr * CT;
        d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int sq = 0; sq < SK; ++sq) {
          const int kr = 4 * sq + fk;
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(El[kr * ks + 16 * tr + fl], F1[(16 * tc + fl) * ks + kr], acc, 0, 0, 0);
        }
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int r = 16 * tr + fk + 4 * rg, c = 16 * tc + fl;
          if (r < K && c < NF) GE[c * ks + r] = acc[rg];
        }
      }
      release_row(il - 1);
      if (il == 8 && tid == 64 && A.ts) A.ts[(4 + w) * 64 + 47] = (double)wall_clock64();
    }
    if (il == 8 && tid == 0 && A.ts) A.ts[(4 + w) * 64 + 48] = (double)wall_clock64();
    __syncthreads();
    nd_ts(A, 4 + w, 9 + 2 * il);
    if (il + 1 < nloc && !prefetched) { wait_row(il + 1); issue(il + 1); }
  }
  if (wave >= 1) {
    publish(nloc - 1, 64, 192);
    release_row(nloc - 1);
  }
  __syncthreads();
  nd_ts(A, 4 + w, 1);
}

// doubles of the separator's Q: per spike workgroup the lower-triangle tiles of Q as the matrix cores leave them
// ([tile][register][lane]: stored and summed without a condition or a transposed copy)
__host__ __device__ constexpr int nd_sep_q_tiles(int K) { return ((2 * K + 1 + 15) / 16) * ((2 * K + 1 + 15) / 16 + 1) / 2; }
__host__ __device__ constexpr int nd_sep_q_doubles(int K) { return 2 * nd_sep_q_tiles(K) * 256; }
// (the whole carve-up of nd_separator)
__host__ __device__ inline int nd_sep_lds_doubles(int K) {
  const int ks = ldl_ks(K);
  return nd_sep_q_doubles(K) + (2 * K + 1) * ks + (K + 1) * ks + 2 * K * ks + K * ks + 4 * ks + 2;
}

// ---- separator workgroup
// While the chains run: wavefronts 0, 1 accumulate Q of spike workgroup 0, wavefronts 2, 3 of spike
// workgroup 1 - each wavefront on its own (no workgroup barrier, MFMA operands straight from the
// rows the spike workgroups publish) - with [Ft | rt] as one block of 2K + 1 columns, so that
//   Q[c][c'] = sum_il Ft_il[:, c] . Dn_il Ft_il[:, c']   and   Q[2K][c] = qy[c] = sum_il rt_il . Dn_il Ft_il[:, c].
// Then: separator system, its two block rows eliminated in registers, back substitution, x_s and
// x_{s+1} published.
// ALT (the seven-workgroup kernel): the two wavefronts of a pair take ALTERNATE rows with all six tiles instead of
// half the tiles of every row.  A row's operands are 24 loads per lane that miss the L2 (the spike workgroup wrote
// them on another XCD), and at K = 23 a wavefront that loads every row took longer per row than the spike
// workgroup needs to produce one: Q was ready 8.8 us after the last row instead of ~3 (allegro N = 60).
template <int K, bool PADDED, bool ALT = false>
__device__ __forceinline__ void nd_separator(const NdArgs& A) {
  extern __shared__ double lds[];
  constexpr int ks = ldl_ks(K), NF = 2 * K, NC = NF + 1, CT2 = (NC + 15) / 16, NQT = CT2 * (CT2 + 1) / 2, QS = NC * NC;
  constexpr int SKq = (K + 3) / 4;
  using d4q = __attribute__((ext_vector_type(4))) double;
  if (threadIdx.x >= 256) return;    // (penta_pipe_kernel launches 512-thread workgroups: four wavefronts work here)
  const int tid = threadIdx.x, nt = 256, lane = tid & 63, wave = tid >> 6;
  const int k = A.k, kk = k * k, s = A.s;
  const NdBuf B = nd_layout(K);
  constexpr int QTS = NQT * 256;     // (nd_sep_q_doubles(K) = 2 QTS)
  double* Qt = lds;                  // [2][NQT][4][64]: the lower-triangle tiles of Q, register rg of lane l of tile t at (4 t + rg) 64 + l
  double* W = Qt + 2 * QTS;          // [2K + 1][ks] columns [S | H | y]
  double* S1 = W + (2 * K + 1) * ks; // [K + 1][ks] second row: [S' | y']
  double* Us = S1 + (K + 1) * ks;    // [2][K][ks] U columns of both rows
  double* Ht = Us + 2 * K * ks;      // [K][ks]
  double* rtv = Ht + K * ks;         // [2][ks]
  double* dnv = rtv + 2 * ks;        // [2][ks]
  nd_ts(A, 6, 0);
  // (rows s, s + 1 assembled by this very launch: wait for their eight words, then read past the L2)
  if (A.asm_ready && tid < 8)
    spin_wait([&] { return __hip_atomic_load(A.asm_ready + 4 * (s + A.asm_first) + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == A.epoch; }, A.spin);
  if (A.asm_ready) __syncthreads();
  auto band = [&](const double* p) { return A.asm_ready ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p; };
  // band blocks and right-hand sides of the separator rows first (no dependence on the chains)
  for (int idx = tid; idx < K * K; idx += nt) {
    const int c = idx / K, r = idx - c * K;
    const bool in = !PADDED || (r < k && c < k);
    W[c * ks + r] = in ? band(A.HC + (size_t)s * kk + c * k + r) : (r == c ? 1.0 : 0.0);
    W[(K + c) * ks + r] = in ? band(A.HB + (size_t)(s + 1) * kk + r * k + c) : 0.0;   // block (s, s+1) = B_{s+1}^T
    S1[c * ks + r] = in ? band(A.HC + (size_t)(s + 1) * kk + c * k + r) : (r == c ? 1.0 : 0.0);
  }
  // (pad rows of the Ht columns are read by the MFMA k-steps of S': LDS is not cleared between
  // kernels, and 0 * NaN from a previous kernel's bits would poison every output)
  for (int idx = tid; idx < K * ks; idx += nt) Ht[idx] = 0.0;
  for (int r = tid; r < K; r += nt) {
    W[2 * K * ks + r] = (r < k) ? A.rhs_sign * band(A.b + (size_t)s * k + r) : 0.0;
    S1[K * ks + r] = (r < k) ? A.rhs_sign * band(A.b + (size_t)(s + 1) * k + r) : 0.0;
  }
  {
    const int w = wave >> 1, sub = wave & 1;            // spike workgroup followed, and which half of its tiles
    const bool mirror = (w == 0);
    const int base = mirror ? A.s - 1 : A.s + 2;
    const int nloc = mirror ? A.s - A.j1 : A.j2 + 2 - (A.s + 2);
    const unsigned long long* frow = A.rowcnt + (2 + w) * ND_MAXROWS;
    const double* Fst = A.ndbuf + B.fst + (size_t)w * ND_MAXROWS * B.frow;
    const unsigned long long ftarget = (A.rowtarget / A.rowunit) * 3ull;   // three wavefronts release a row
    const int fl = lane & 15, fk = lane >> 4;
    constexpr int NACC = ALT ? NQT : (NQT + 1) / 2;
    d4q qacc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t) qacc[t] = d4q{0.0, 0.0, 0.0, 0.0};
    for (int il = ALT ? sub : 0; il < nloc; il += ALT ? 2 : 1) {
      spin_wait([&] { return __hip_atomic_load(frow + il, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= ftarget; }, A.spin);
      if (A.wt_rows != 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // (no second round trip for an acquiring load: the poll has seen the count)
      if (A.ts && lane == 0 && w == 0 && il < 24) A.ts[6 * 64 + 24 + il] = (double)wall_clock64();   // (debug: row il of spike workgroup 0 seen)
      const double* F = Fst + (size_t)il * B.frow;
      const double* dg = A.Dst + (size_t)(mirror ? base - il : base + il) * K;
      double op[CT2][SKq], dn[SKq];
      if (A.wt_rows == 1) {
#pragma unroll
        for (int sq = 0; sq < SKq; ++sq) {
          const int kr = 4 * sq + fk;
          dn[sq] = (kr < K) ? __hip_atomic_load(dg + kr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
#pragma unroll
          for (int t = 0; t < CT2; ++t) op[t][sq] = __hip_atomic_load(F + (16 * t + fl) * ks + kr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      } else {
#pragma unroll
        for (int sq = 0; sq < SKq; ++sq) {
          const int kr = 4 * sq + fk;
          dn[sq] = (kr < K) ? dg[kr] : 0.0;
#pragma unroll
          for (int t = 0; t < CT2; ++t) op[t][sq] = F[(16 * t + fl) * ks + kr];
        }
      }
      if (A.ts && w == 0 && il == nloc - 1) {   // (debug: the last row's operands are in registers)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) A.ts[6 * 64 + 50] = (double)wall_clock64();
      }
      // tile t -> wavefront t % 2 of the pair, accumulator t / 2 (ALT: every tile, accumulator t).  The k-steps are the
      // OUTER loop: consecutive matrix-core instructions then go to different accumulators (a tile's own k-steps are
      // a dependent chain: 80 of them took 5.9 us at K = 29), and the scaled operand is formed once per column tile.
#pragma unroll
      for (int sq = 0; sq < SKq; ++sq) {
        double opd[CT2];
#pragma unroll
        for (int t = 0; t < CT2; ++t) opd[t] = op[t][sq] * dn[sq];
        nd_static_for<0, NQT>([&](auto tt) __attribute__((always_inline)) {
          constexpr int t = decltype(tt)::value, tr = nd_tile_row(t), tc = nd_tile_col(t), ai = ALT ? t : t / 2;
          if (ALT || t % 2 == sub) qacc[ai] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[tr][sq], opd[tc], qacc[ai], 0, 0, 0);
        });
      }
    }
    if (A.ts && w == 0 && lane == 0 && sub == ((nloc - 1) & 1)) A.ts[6 * 64 + 51] = (double)wall_clock64();   // (debug: ... and accumulated)
    // (ALT: the wavefront of the pair that does NOT hold the last row stores its sums - it is done a row earlier -, the
    // other one adds its own to them after a barrier)
    const int late = (nloc - 1) & 1;   // the wavefront with the last row
#pragma unroll
    for (int pass = 0; pass < (ALT ? 2 : 1); ++pass) {
      if (ALT && pass == 1) __syncthreads();
#pragma unroll
      for (int t = 0; t < NQT; ++t) {
        if (ALT ? ((sub == late) != (pass == 1)) : (t % 2 != sub)) continue;
        const int ai = ALT ? t : t / 2;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          double* at = Qt + w * QTS + (4 * t + rg) * 64 + lane;
          *at = (ALT && pass == 1) ? *at + qacc[ai][rg] : qacc[ai][rg];
        }
      }
    }
  }
  __syncthreads();
  nd_ts(A, 6, 1);
  // column blocks of the two spike workgroups: w = 0 (mirrored chain J1): nearest = s (offset 0), farthest = s+1
  // (offset K); w = 1 (chain J2): nearest = s+1, farthest = s; column 2K is rt
  // element (R, C) of the symmetric Q of spike workgroup w (the lower triangle is what the tiles hold: a diagonal tile
  // has both (R, C) and (C, R), summed in different orders)
  auto qat = [&](int w, int R, int C) {
    const int hi = R > C ? R : C, lo = R > C ? C : R;
    const int tr = hi >> 4, tc = lo >> 4, rr = hi & 15;
    return Qt[w * QTS + (4 * (tr * (tr + 1) / 2 + tc) + (rr >> 2)) * 64 + (rr & 3) * 16 + (lo & 15)];
  };
  auto qs = [&](int w, int rowsel, int colsel, int r, int c) {   // rowsel / colsel: 0 -> block of row s, 1 -> of s+1
    const int ro = (w == 0 ? rowsel : 1 - rowsel) * K, co = (w == 0 ? colsel : 1 - colsel) * K;
    return qat(w, ro + r, co + c);
  };
  auto qyv = [&](int w, int rowsel, int r) { return qat(w, NF, (w == 0 ? rowsel : 1 - rowsel) * K + r); };
  for (int idx = tid; idx < K * K; idx += nt) {
    const int c = idx / K, r = idx - c * K;
    W[c * ks + r] = (W[c * ks + r] - qs(0, 0, 0, r, c)) - qs(1, 0, 0, r, c);
    W[(K + c) * ks + r] = (W[(K + c) * ks + r] - qs(0, 0, 1, r, c)) - qs(1, 0, 1, r, c);
    S1[c * ks + r] = (S1[c * ks + r] - qs(0, 1, 1, r, c)) - qs(1, 1, 1, r, c);
  }
  for (int r = tid; r < K; r += nt) {
    W[2 * K * ks + r] = (W[2 * K * ks + r] - qyv(0, 0, r)) - qyv(1, 0, r);
    S1[K * ks + r] = (S1[K * ks + r] - qyv(0, 1, r)) - qyv(1, 1, r);
  }
  __syncthreads();
  nd_ts(A, 6, 3);
  bool bad = false;
  // ---- row s: [S | H | y] in the registers of wavefront 0
  if (wave == 0) {
    const int col = (lane < 2 * K + 1) ? lane : 0;
    double xr[K];
#pragma unroll
    for (int r = 0; r < K; ++r) xr[r] = W[col * ks + r];
    const double diag0 = (lane < K) ? W[lane * ks + lane] : 1.0;
    double yacc;
    const double myinv = ldl_eliminate_wave<K, false>(xr, lane, yacc);
    bad = bad || (lane < K && ldl_pivot_bad(myinv, diag0, lane, k, A.npos));
    if (lane < K) {
#pragma unroll
      for (int r = 0; r < K; ++r) Us[lane * ks + r] = xr[r];
      dnv[lane] = myinv;
      A.Dst[(size_t)s * K + lane] = myinv;   // (kkt_extract_kernel looks at every multiplier pivot)
    } else if (lane < 2 * K) {
#pragma unroll
      for (int r = 0; r < K; ++r) Ht[(lane - K) * ks + r] = xr[r];
    } else if (lane == 2 * K) {
#pragma unroll
      for (int r = 0; r < K; ++r) rtv[r] = xr[r];
    }
  }
  __syncthreads();
  nd_ts(A, 6, 4);
  // ---- row s+1: S' = S_{s+1} - Ht^T Dn Ht (matrix cores, lower tiles mirrored), y' = y_{s+1} - Ht^T Dn rt
  {
    using d4 = __attribute__((ext_vector_type(4))) double;
    constexpr int SK = (K + 3) / 4, TT = (K + 15) / 16, NT2 = TT * (TT + 1) / 2;
    const int fl = lane & 15, fk = lane >> 4;
    if (wave < NT2) {
      const int tr = nd_tile_row(wave), tc = nd_tile_col(wave);
      d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int sq = 0; sq < SK; ++sq) {
        const int kr = 4 * sq + fk;
        const double dnk = (kr < K) ? dnv[kr] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ht[(16 * tr + fl) * ks + kr], Ht[(16 * tc + fl) * ks + kr] * dnk, acc, 0, 0, 0);
      }
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int r = 16 * tr + fk + 4 * rg, c = 16 * tc + fl;
        if (r < K && c < K && r >= c) {
          const double val = S1[c * ks + r] - acc[rg];
          S1[c * ks + r] = val;
          if (r != c) S1[r * ks + c] = val;
        }
      }
    } else if (wave == NT2 && lane < K) {   // (TT = 2: three tiles on wavefronts 0..2, the right-hand side on wavefront 3)
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < K; ++j) acc = __builtin_fma(Ht[lane * ks + j] * dnv[j], rtv[j], acc);
      S1[K * ks + lane] -= acc;
    }
    static_assert(NT2 <= 3, "separator: S' tiles are spread over three wavefronts");
  }
  __syncthreads();
  nd_ts(A, 6, 5);
  if (wave == 0) {
    const int col = (lane < K + 1) ? lane : 0;
    double xr[K];
#pragma unroll
    for (int r = 0; r < K; ++r) xr[r] = S1[col * ks + r];
    const double diag0 = (lane < K) ? S1[lane * ks + lane] : 1.0;
    double yacc;
    const double myinv = ldl_eliminate_wave<K, false>(xr, lane, yacc);
    bad = bad || (lane < K && ldl_pivot_bad(myinv, diag0, lane, k, A.npos));
    if (lane < K) {
#pragma unroll
      for (int r = 0; r < K; ++r) Us[(K + lane) * ks + r] = xr[r];
      dnv[ks + lane] = myinv;
      A.Dst[(size_t)(s + 1) * K + lane] = myinv;
    } else if (lane == K) {
#pragma unroll
      for (int r = 0; r < K; ++r) rtv[ks + r] = xr[r];
    }
    if (__builtin_amdgcn_ballot_w64(bad) != 0ull && lane == 0) {
      __hip_atomic_store(A.status, A.fact_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_fetch_add(A.status + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (A.ts) A.ts[6 * 64 + 20] = 1.0;
    }
  }
  __syncthreads();
  nd_ts(A, 6, 6);
  // ---- back substitution (U = D L^T: x_r = rt_r / d_r - sum_{c > r} (U[r][c] / d_r) x_c), lane = row
  if (wave == 0) {
    const int r = (lane < K) ? lane : 0;
    // rows r of D^-1 U_{s+1}, D^-1 U_s (strict upper parts) and of Ht, fetched before the dependent chains start
    double u1[K], u0[K], hrow[K];
    const double d1 = dnv[ks + r], d0 = dnv[r];
#pragma unroll
    for (int j = 0; j < K; ++j) {
      u1[j] = Us[(K + j) * ks + r];
      u0[j] = Us[j * ks + r];
      hrow[j] = Ht[j * ks + r];
    }
    double v = rtv[ks + r] * d1;
#pragma unroll
    for (int j = K - 1; j >= 0; --j) v = __builtin_fma(-((r < j) ? u1[j] * d1 : 0.0), rdlane(v, j), v);
    const double x1 = v;                                         // x_{s+1}
    double acc = 0.0;
#pragma unroll
    for (int c = 0; c < K; ++c) acc = __builtin_fma(hrow[c], rdlane(x1, c), acc);   // (Ht x_{s+1})[r]
    v = (rtv[r] - acc) * d0;
#pragma unroll
    for (int j = K - 1; j >= 0; --j) v = __builtin_fma(-((r < j) ? u0[j] * d0 : 0.0), rdlane(v, j), v);
    const double x0 = v;                                         // x_s
    if (lane < K) {
      double* xsep = A.ndbuf + B.xsep;
      xsep[lane] = x0;
      xsep[K + lane] = x1;
      double* xll = A.ndbuf + B.ll;
      ll_store(xll + 2 * lane, x0, A.epoch);
      ll_store(xll + 2 * (K + lane), x1, A.epoch);
      if (lane < k) { A.x[(size_t)s * k + lane] = x0; A.x[(size_t)(s + 1) * k + lane] = x1; }
    }
  }
  nd_post(A.flags + 4, A.epoch);
  nd_ts(A, 6, 2);
}

// grid (7, batch): blockIdx.x = role, blockIdx.y = problem of the batch
//   0: P0 producer, rows 0 .. j1-1 top-down        1: P3 producer, rows n-1 .. j2+2 bottom-up
//   2: J1 joiner, rows s-1 .. j1+2 then j1+1, j1   3: J2 joiner, rows s+2 .. j2-1 then j2, j2+1
//   4, 5: spike workgroups of J1, J2               6: separator
// (workgroups are dispatched in this order: every wait is on a lower index or on the partner two
// places up, see fused.h for the forward-progress argument)
template <int K, bool PADDED>
__global__ void __launch_bounds__(256) penta_nd_kernel(NdArgs A) {
  {
    const size_t o = (size_t)blockIdx.y * A.pstride;
    A.HA = at_problem(A.HA, o); A.HB = at_problem(A.HB, o); A.HC = at_problem(A.HC, o); A.b = at_problem(A.b, o);
    A.x = at_problem(A.x, o); A.Ust = at_problem(A.Ust, o); A.Hst = at_problem(A.Hst, o); A.Est = at_problem(A.Est, o);
    A.Dst = at_problem(A.Dst, o); A.xch = at_problem(A.xch, o); A.flags = at_problem(A.flags, o);
    A.rowcnt = at_problem(A.rowcnt, o); A.ndbuf = at_problem(A.ndbuf, o);
    if (A.wst) A.wst = at_problem(A.wst, o);
    A.spin = SpinCtl{A.status + 2 * gridDim.y, A.fact_id};
    A.status += 2 * blockIdx.y;
  }
  const int role = blockIdx.x;
  if (role == A.debug_skip_role) return;
  if (role == 6) { nd_separator<K, PADDED, true>(A); return; }
  if (role >= 4) { nd_spike<K, PADDED>(A, role - 4); return; }
  const NdBuf B = nd_layout(K);
  ChainCfg c = {};
  c.two = 1;
  c.dbg_slot = role;
  c.ts = A.ts ? A.ts + role * 64 : nullptr;
  c.spin = A.spin;
  c.npos = A.npos; c.lds_rows = A.lds_rows;
  int pair;
  if (role == 0) { c.mirror = 0; c.producer = 1; c.base = 0; c.nloc = A.j1; pair = 0; }
  else if (role == 1) { c.mirror = 1; c.producer = 1; c.base = A.n - 1; c.nloc = A.n - A.j2 - 2; pair = 1; }
  else {
    const int w = role - 2;
    c.mirror = (w == 0); c.producer = 0;
    c.base = (w == 0) ? A.s - 1 : A.s + 2;
    c.m_split = (w == 0) ? A.s - A.j1 - 2 : A.j2 - A.s - 2;
    c.nloc = c.m_split + 2;
    pair = w;
    c.rowcnt = A.rowcnt + w * ND_MAXROWS; c.rowcnt_unit = 1ull;
    c.rtpub = A.ndbuf + B.rtpub + (size_t)w * ND_MAXROWS * K;
    c.fst = A.ndbuf + B.fst + (size_t)w * ND_MAXROWS * B.frow; c.fstride = B.frow;
    c.frowcnt = A.rowcnt + (2 + w) * ND_MAXROWS; c.frowtarget = (A.rowtarget / A.rowunit) * 3ull;
    c.xsep = A.ndbuf + B.xsep; c.sepflag = A.flags + 4;
    c.xsep_ll = A.ndbuf + B.ll;
    c.wst = A.wst ? A.wst + (size_t)w * ND_MAXROWS * B.frow : nullptr;
  }
  c.rec_tail = A.rec_tail && A.wst; c.lds_doubles = A.lds_doubles;
  c.xjoin_ll = A.ndbuf + B.ll + (1 + pair) * 4 * K;
  // (one word per row of W, behind what the pair's exchange buffer holds: two augmented blocks and the join rows of x)
  c.wrow = reinterpret_cast<unsigned*>(A.xch + (size_t)pair * A.xch_pair + ((2 * (3 * K + 1) * ldl_ks(K) + 2 * K + 1) & ~1));
  if (c.rec_tail && role < 2) {       // the joiner of this producer's pair
    c.wp_mirror = (pair == 0); c.wp_base = (pair == 0) ? A.s - 1 : A.s + 2;
    c.wp_nloc = (pair == 0) ? A.s - A.j1 : A.j2 - A.s;
    c.wp_fst = A.ndbuf + B.fst + (size_t)pair * ND_MAXROWS * B.frow; c.fstride = B.frow;
    c.wp_frowcnt = A.rowcnt + (2 + pair) * ND_MAXROWS; c.frowtarget = (A.rowtarget / A.rowunit) * 3ull;
    c.wp_wst = A.wst + (size_t)pair * ND_MAXROWS * B.frow;
    c.wp_waves = A.rec_tail;
  }
  constexpr int GW = ((2 * K + 1) + (64 - K) - 1) / (64 - K);
  penta_ldl_body<K, 256, PADDED, GW, (K > 20)>(A.n, A.k, A.HA, A.HB, A.HC, A.b, A.rhs_sign, 1, A.x, A.Ust, A.Hst, A.Est, A.Dst,
                                     nullptr, c, A.xch + (size_t)pair * A.xch_pair, A.flags + 2 * pair, A.epoch, A.status,
                                     A.fact_id);
}

}  // namespace idto_dev
