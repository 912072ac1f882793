// fused.h — one persistent launch per Gauss-Newton iteration (single-problem contexts).
//
// fd_kernel -> assemble_diag_kernel -> penta_ldl_kernel are dependent launches of 40 / 164 / 2
// workgroups: on a 256-CU device all of them fit at once, so the three kernels become ROLES of one
// grid and the stream-order dependencies become two monotonic counters in device memory:
//   blocks [0, nfd)              finite-difference block k        -> ++sync[0] when its record is out
//   blocks [nfd, nfd + nasm)     assembly (block row i, part p)   wait sync[0] == all fd blocks; ++sync[1]
//   blocks [nfd + nasm, ...)     the solver's workgroups          wait sync[1] == all assembly blocks
// No launch gaps (two dependent-launch latencies per iteration saved), the assembly and solver
// workgroups are resident and have staged their constant operands before their inputs arrive.
// Forward progress does not need a cooperative launch: workgroups are dispatched in block-index
// order and every wait is on workgroups with a LOWER index (already dispatched; they never wait on
// higher ones) - except the two solver sides, which wait on each other and are adjacent in the
// index order (the second is dispatched as soon as a slot frees up, which the waiting first one
// does not prevent: every other resident workgroup finishes on its own).
// Release / acquire: agent scope (the roles run on different XCDs, each with its own L2).
#pragma once

#include "kernels.h"
#include "penta_ldl.h"

namespace idto_dev {

struct FusedArgs {
  DevModel M;
  DevContact cp;
  DevProblem P;
  // finite differences
  double* q; double* slab; int slab_stride; double* v; double* a; double* nplus;
  int fd_mode, fd_echunk, nfd;
  // assembly
  double* g; double* HA; double* HB; double* HC;
  int nrows;  // N + 1 block rows, 4 parts each
  // solver (sub-system starting at block row r0: pointers already advanced)
  int n, k;
  double* sHA; double* sHB; double* sHC; double* b; double rhs_sign; double* x;
  double* Ust; double* Hst; double* Est; double* Dst;
  int m_split; double* xch; unsigned* flags; unsigned epoch; unsigned* status; unsigned fact_id;
  // synchronisation
  unsigned long long* sync; unsigned long long fd_target, asm_target;
  double* dbg;  // option "solver_debug": per workgroup [start, inputs ready / body done, end] in 100 MHz ticks
};

// Release: the workgroup barrier orders every wavefront's global writes before thread 0's
// agent-scope release (ONE L2 write-back per workgroup; a __threadfence() in every thread made
// each of the four wavefronts write the L2 back and invalidate it: 164 assembly workgroups doing
// that at once stretched the phase from 6 to 26 us).
__device__ __forceinline__ void fused_signal(unsigned long long* cnt) {
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// Polling is RELAXED (a load that reaches the agent-coherent level, no cache invalidation per
// poll) with `nap` x 64 cycles between polls; one acquire, then the barrier publishes it to the
// other wavefronts of the workgroup (they share the CU's vector L1, which the acquire invalidated).
__device__ __forceinline__ void fused_wait(unsigned long long* cnt, unsigned long long target, bool many, const SpinCtl sc) {
  if (threadIdx.x == 0) {
    // (bounded: penta_ldl.h spin_wait; `many` pollers nap longer between polls)
    spin_wait([&] {
      if (many) __builtin_amdgcn_s_sleep(7);
      return __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target;
    }, sc);
    (void)__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
}
__device__ __forceinline__ void fused_stamp(double* dbg, int slot) {
  if (dbg && threadIdx.x == 0) dbg[blockIdx.x * 4 + slot] = (double)wall_clock64();
}

// (single-problem launch: the word a wait that ran out reports to sits behind the problem's two status words)
__device__ __forceinline__ ChainCfg fused_chain_cfg(const FusedArgs& A, int side) {
  ChainCfg c = two_sided_cfg(A.n, A.m_split, side);
  c.spin = SpinCtl{A.status + 2, A.fact_id};
  return c;
}

template <int MAXC, int K, bool PADDED, int GJW>
__global__ void __launch_bounds__(256) gn_fused_kernel(FusedArgs A) {
  const int bx = blockIdx.x;
  const int nasm = 4 * A.nrows;
  fused_stamp(A.dbg, 0);
  if (bx < A.nfd) {
    fd_body<MAXC>(A.M, A.cp, A.P, A.q, A.slab, A.slab_stride, A.v, A.a, A.nplus, bx, A.fd_mode, 0, A.fd_echunk, nullptr);
    fused_stamp(A.dbg, 1);
    fused_signal(A.sync);
    fused_stamp(A.dbg, 2);
  } else if (bx < A.nfd + nasm) {
    const int idx = bx - A.nfd;
    fused_wait(A.sync, A.fd_target, true, SpinCtl{A.status + 2, A.fact_id});
    fused_stamp(A.dbg, 1);
    assemble_diag_body(A.M, A.P, A.q, A.slab, A.slab_stride, A.g, A.HA, A.HB, A.HC, 0, A.v, A.nplus, idx >> 2, idx & 3);
    fused_signal(A.sync + 1);
    fused_stamp(A.dbg, 2);
  } else {
    fused_wait(A.sync + 1, A.asm_target, false, SpinCtl{A.status + 2, A.fact_id});
    fused_stamp(A.dbg, 1);
    penta_ldl_body<K, 256, PADDED, GJW>(A.n, A.k, A.sHA, A.sHB, A.sHC, A.b, A.rhs_sign, 1, A.x, A.Ust, A.Hst, A.Est,
                                        A.Dst, nullptr, fused_chain_cfg(A, bx - A.nfd - nasm), A.xch, A.flags,
                                        A.epoch, A.status, A.fact_id);
    fused_stamp(A.dbg, 2);
  }
}

}  // namespace idto_dev
