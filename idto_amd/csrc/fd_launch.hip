// fd_launch.hip — the instantiations of fd_kernel (fd_kernel.h) and their launch.
#include "fd_launch.h"

namespace idto_dev {

#define FD_ARGS a.M, a.cp, a.P, a.q, a.slab, a.slab_stride, a.v, a.a, a.nplus, a.k_begin, a.mode, a.stop_after, a.echunk, \
                a.pstride, a.terms, a.alt
#define FD_GO(MC, SH) hipLaunchKernelGGL((fd_kernel<MC, SH>), a.grid, a.block, a.lds, a.stream, FD_ARGS)

void fd_launch(const FdLaunch& a) {
  if (a.shape == 1) FD_GO(2, 1);
  else if (a.shape == 2) FD_GO(3, 2);
  else if (a.shape == 3) FD_GO(3, 3);
  else if (a.shape == 4) FD_GO(4, 4);
  else if (a.shape == 5) FD_GO(3, 5);
  else if (a.maxc <= 2) FD_GO(2, 0);
  else if (a.maxc <= 3) FD_GO(3, 0);
  else if (a.maxc <= 4) FD_GO(4, 0);
  else FD_GO(8, 0);
}

#define FD_ATTR(MC, SH) \
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fd_kernel<MC, SH>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds)

void fd_set_max_lds(int max_lds) {
  FD_ATTR(2, 0); FD_ATTR(3, 0); FD_ATTR(4, 0); FD_ATTR(8, 0);
  FD_ATTR(2, 1); FD_ATTR(3, 2); FD_ATTR(3, 3); FD_ATTR(4, 4); FD_ATTR(3, 5);
}

}  // namespace idto_dev
