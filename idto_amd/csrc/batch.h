// batch.h — addressing of batched contexts.  Every per-problem array of a context lives in one
// arena per problem; all arenas have the same layout and lie `pstride` BYTES apart
// (idto_hip.hip).  A kernel instance working on problem b (blockIdx.y; .z for the assembly)
// shifts the problem-0 pointers it was given by b * pstride.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

namespace idto_dev {

template <class T>
__host__ __device__ __forceinline__ T* at_problem(T* p, size_t byte_off) {
  using U = typename std::remove_const<T>::type;
  return reinterpret_cast<T*>(reinterpret_cast<char*>(const_cast<U*>(p)) + byte_off);
}

// Two sets of what fd_kernel writes (v, a, N+, the slab of partials and tau, the assembly products)
// inside one arena, `off` bytes apart (idto_hip_tr_solve): the device-resident trust-region loop
// evaluates the trial point into the set the iterate does NOT occupy, and accepting a step is
// flipping one word of device state - a rejected step keeps the iterate's partials.  `which`: 0 the
// iterate's set, 1 the other one.  state == nullptr: set A, always.
constexpr int IDTO_TRS_CUR = 9;   // index of the current-set word in the loop's device state (trust_region.h TRS_CUR)
struct AltSel {
  const double* state;
  long long off;
  int which;
};
// (`o`: the problem's arena offset in bytes - every problem of a batch has its own loop state, hence its own current set)
__device__ __forceinline__ long long alt_offset(const AltSel& s, size_t o = 0) {
  if (!s.state) return 0;
  const double* st = reinterpret_cast<const double*>(reinterpret_cast<const char*>(s.state) + o);
  const bool cur1 = st[IDTO_TRS_CUR] != 0.0;
  return (cur1 != (s.which != 0)) ? s.off : 0;
}
template <class T>
__device__ __forceinline__ T* at_set(T* p, const AltSel& s) {
  using U = typename std::remove_const<T>::type;
  return reinterpret_cast<T*>(reinterpret_cast<char*>(const_cast<U*>(p)) + alt_offset(s));
}

}  // namespace idto_dev
