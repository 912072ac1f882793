// Microbenchmark: cost of back-to-back LDS instructions issued by ONE wavefront of a workgroup
// (the solver's situation), in cycles per instruction.  hipcc --offload-arch=gfx950 -O3 lds_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double __attribute__((ext_vector_type(2))) v2d;
__global__ void __launch_bounds__(256) kg(long long* out, const double* g, double* gw, int active_waves) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave >= active_waves) return;
  v2d acc = {0, 0};
  const v2d* p = (const v2d*)(g + wave * 8192 + lane * 22);
  long long t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < 32; ++i) acc += p[i * 1024 / 16 * 0 + i];   // 32 independent global_load_dwordx4, L2-warm
  long long t1 = __builtin_readcyclecounter();
  v2d* q = (v2d*)(gw + wave * 8192 + lane * 2);
#pragma unroll
  for (int i = 0; i < 32; ++i) q[i * 64] = acc;                    // 32 global_store_dwordx4 (fire and forget)
  long long t2 = __builtin_readcyclecounter();
  if (lane == 0) { out[wave] = t1 - t0; out[4 + wave] = t2 - t1; }
  if (acc.x == 12345.678) out[15] = 1;
}
template <int MODE>
__global__ void __launch_bounds__(256) k(long long* out, int stride_doubles, int active_waves) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = i;
  __syncthreads();
  if (wave >= active_waves) return;
  double acc = 0;
  v2d a2 = {0, 0};
  const double* p = lds + wave * 2048 + lane * stride_doubles;
  long long t0 = __builtin_readcyclecounter();
  if (MODE == 0) {  // 32 independent ds_read_b128
#pragma unroll
    for (int i = 0; i < 32; ++i) a2 += *(const v2d*)(p + 2 * i + ((i & 1) ? 64 : 0));
  } else if (MODE == 1) {  // 32 independent ds_read_b64
#pragma unroll
    for (int i = 0; i < 32; ++i) acc += p[i];
  } else if (MODE == 2) {  // 32 ds_write_b128
#pragma unroll
    for (int i = 0; i < 32; ++i) *(v2d*)(lds + wave * 2048 + lane * stride_doubles + 2 * i) = (v2d){(double)i, (double)lane};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  } else {  // 32 dependent (pointer-chasing) ds_read_b64: latency
    int idx = lane;
#pragma unroll
    for (int i = 0; i < 32; ++i) idx = (int)lds[idx & 1023] & 1023;
    acc = idx;
  }
  acc += a2.x + a2.y;
  long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[wave] = t1 - t0;
  if (acc == 12345.678) out[8] = 1;
}
int main() {
  long long* d; hipMalloc(&d, 128);
  long long h[16];
  const char* names[4] = {"32 x ds_read_b128 (independent)", "32 x ds_read_b64 (independent)", "32 x ds_write_b128", "32 x dependent ds_read_b64"};
  for (int mode = 0; mode < 4; ++mode)
    for (int waves : {1, 4})
      for (int stride : {2, 22}) {
        for (int rep = 0; rep < 3; ++rep) {
          if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(256), 65536, 0, d, stride, waves);
          if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(256), 65536, 0, d, stride, waves);
          if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(256), 65536, 0, d, stride, waves);
          if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(256), 65536, 0, d, stride, waves);
          hipDeviceSynchronize();
        }
        hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
        printf("%-36s waves=%d lane stride=%2d doubles: %6.1f cycles/instr (wave 0)\n", names[mode], waves, stride, h[0] / 32.0);
      }
  double *g, *gw; hipMalloc(&g, 8 * 8192 * 8); hipMalloc(&gw, 8 * 8192 * 8); hipMemset(g, 0, 8 * 8192 * 8);
  for (int waves : {1, 4}) {
    for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(kg, dim3(1), dim3(256), 0, 0, d, g, gw, waves); hipDeviceSynchronize(); }
    hipMemcpy(h, d, 128, hipMemcpyDeviceToHost);
    printf("32 x global_load_dwordx4 (independent, issue + return) waves=%d: %6.1f cycles/instr;  32 x global_store_dwordx4: %6.1f cycles/instr\n",
           waves, h[0] / 32.0, h[4] / 32.0);
  }
  return 0;
}
