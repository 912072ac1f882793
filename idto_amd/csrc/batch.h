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

}  // namespace idto_dev
