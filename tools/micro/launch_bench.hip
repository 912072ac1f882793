// launch_bench.hip — what an (almost) empty kernel costs as a function of how it is launched: dynamic LDS, registers
// per lane, scratch, size of the kernel arguments, grid.  Back-to-back launches on one stream, wall time per launch.
// hipcc --offload-arch=gfx950 -O3 tools/micro/launch_bench.hip -o tools/micro/launch_bench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

struct Big { double v[64]; };   // 512 bytes of kernel arguments

__global__ void k_small(double* out) { if (out && threadIdx.x == 9999) out[0] = 1; }
__global__ void k_bigarg(Big b, double* out) { if (out && threadIdx.x == 9999) out[0] = b.v[blockIdx.x & 63]; }
__global__ void __launch_bounds__(256) k_regs(double* out, int n) {   // ~500 registers per lane, used only when n != 0
  extern __shared__ double lds[];
  if (n == 0) return;
  double acc[240];
#pragma unroll
  for (int i = 0; i < 240; ++i) acc[i] = lds[(threadIdx.x + i) & 1023];
  for (int it = 0; it < n; ++it)
#pragma unroll
    for (int i = 0; i < 240; ++i) acc[i] = acc[i] * 1.0000001 + acc[(i + 7) % 240];
  double s = 0;
#pragma unroll
  for (int i = 0; i < 240; ++i) s += acc[i];
  out[threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_scratch(double* out, int n) {   // a dynamically indexed private array: scratch
  if (n == 0) return;
  double a[64];
  for (int i = 0; i < 64; ++i) a[i] = i * out[0];
  double s = 0;
  for (int i = 0; i < n; ++i) s += a[(i * 7 + threadIdx.x) & 63];
  out[threadIdx.x] = s;
}

template <class F>
static double time_us(F launch, int reps = 2000) {
  for (int i = 0; i < 200; ++i) launch();
  hipDeviceSynchronize();
  auto t0 = std::chrono::high_resolution_clock::now();
  for (int i = 0; i < reps; ++i) launch();
  hipDeviceSynchronize();
  return std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / reps;
}

int main() {
  double* out;
  hipMalloc(&out, 1 << 20);
  hipMemset(out, 0, 1 << 20);
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_regs), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_small), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  Big b{};
  printf("back-to-back launches on one stream, us per launch (1x MI355X)\n");
  for (int grid : {1, 40, 169}) {
    printf("grid %3d x 256 threads:\n", grid);
    printf("  small kernel, no LDS                         %6.2f\n", time_us([&] { hipLaunchKernelGGL(k_small, dim3(grid), dim3(256), 0, st, out); }));
    printf("  small kernel, 64 KB dynamic LDS              %6.2f\n", time_us([&] { hipLaunchKernelGGL(k_small, dim3(grid), dim3(256), 64 * 1024, st, out); }));
    printf("  small kernel, 160 KB dynamic LDS             %6.2f\n", time_us([&] { hipLaunchKernelGGL(k_small, dim3(grid), dim3(256), 160 * 1024, st, out); }));
    printf("  512 B of kernel arguments                    %6.2f\n", time_us([&] { hipLaunchKernelGGL(k_bigarg, dim3(grid), dim3(256), 0, st, b, out); }));
    printf("  ~500 registers per lane (returns at once)    %6.2f\n", time_us([&] { hipLaunchKernelGGL(k_regs, dim3(grid), dim3(256), 8192, st, out, 0); }));
    printf("  ~500 registers, 160 KB LDS (returns at once) %6.2f\n", time_us([&] { hipLaunchKernelGGL(k_regs, dim3(grid), dim3(256), 160 * 1024, st, out, 0); }));
    printf("  private array in scratch (returns at once)   %6.2f\n", time_us([&] { hipLaunchKernelGGL(k_scratch, dim3(grid), dim3(256), 0, st, out, 0); }));
  }
  return 0;
}
