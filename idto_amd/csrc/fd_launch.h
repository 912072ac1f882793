// fd_launch.h — how idto_hip.hip starts fd_kernel: its instantiations live in csrc/fd_launch.hip, a
// translation unit of their own (built in parallel with the solvers, with its own scheduling flags: build.sh).
#pragma once

#include <hip/hip_runtime.h>

#include "fd_kernel.h"

namespace idto_dev {

struct FdLaunch {
  dim3 grid, block;
  int lds;
  hipStream_t stream;
  DevModel M;
  DevContact cp;
  DevProblem P;
  const double* q;
  double* slab;
  int slab_stride;
  double *v, *a, *nplus;
  int k_begin, mode, stop_after, echunk;
  size_t pstride;
  double* terms;
  AltSel alt;
  int shape;   // id_fast.h: the model's instantiated tree shape, 0: id_eval<MAXC>
  int maxc;
};
// enqueues fd_kernel<MAXC, SHAPE>; the caller checks hipGetLastError()
void fd_launch(const FdLaunch& a);
// opt-in of every instantiation to `max_lds` bytes of dynamic LDS
void fd_set_max_lds(int max_lds);

}  // namespace idto_dev
