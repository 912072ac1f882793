// Latency of the dependent chain of one pivot of the block LDL^T elimination (penta_pipe.h pipe_pivot):
//   1/d_J -> d_{J+1} = a - u^2 / d_J -> v_readlane -> v_rcp_f64 -> refinement -> 1/d_{J+1}
// one wavefront, cycles (s_memtime) per link, for the pieces and the whole.  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ double rdlane(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rcp_cubic(double d) {
  const double x = __builtin_amdgcn_rcp(d);
  const double e = __builtin_fma(-d, x, 1.0);
  const double p = __builtin_fma(e, e, e);
  return __builtin_fma(x, p, x);
}
__device__ __forceinline__ double rcp_newton2(double d) {
  double x = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, x, 1.0);
  x = __builtin_fma(x, e, x);
  e = __builtin_fma(-d, x, 1.0);
  return __builtin_fma(x, e, x);
}

template <int MODE>
__global__ void __launch_bounds__(64) chain_kernel(long long* out, double* sink, double seed) {
  const int lane = threadIdx.x;
  double a = seed + lane * 1e-3, u = 0.25 + lane * 1e-4, inv = 1.0 / (seed + 1.0), x = seed;
  constexpr int N = 512;
  const long long t0 = __builtin_readcyclecounter();
#pragma unroll 16
  for (int i = 0; i < N; ++i) {
    if (MODE == 0) x = __builtin_fma(x, 0.999, 1e-9);                        // dependent v_fma_f64
    if (MODE == 1) x = __builtin_amdgcn_rcp(x) + 1.5;                         // dependent v_rcp_f64 (+ an add)
    if (MODE == 2) x = rdlane(x, i & 63) + 1e-9;                              // v_readlane pair + an add
    if (MODE == 3) inv = rcp_cubic(inv + 1.5);                                // reciprocal with the cubic step
    if (MODE == 4) inv = rcp_newton2(inv + 1.5);                              // ... with two Newton steps
    if (MODE == 5) {                                                          // the whole link
      const double dn = __builtin_fma(-(u * u), inv, a);
      inv = rcp_cubic(rdlane(dn, (i + 1) & 63));
    }
    if (MODE == 6) {                                                          // the link as penta_ldl.h has it: scaled row, update, reciprocal
      const double t = u * inv;
      const double m1 = rdlane(a, i & 63);
      a = __builtin_fma(-m1, t, a + 1.0);
      inv = rcp_newton2(rdlane(a, (i + 1) & 63));
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[MODE] = (t1 - t0) / N;
  sink[lane + 64 * MODE] = x + inv + a;
}

int main() {
  long long* out; double* sink;
  hipMalloc(&out, 16 * sizeof(long long)); hipMalloc(&sink, 64 * 16 * sizeof(double));
  hipMemset(out, 0, 16 * sizeof(long long));
  for (int rep = 0; rep < 2; ++rep) {
    chain_kernel<0><<<1, 64>>>(out, sink, 1.25); chain_kernel<1><<<1, 64>>>(out, sink, 1.25); chain_kernel<2><<<1, 64>>>(out, sink, 1.25);
    chain_kernel<3><<<1, 64>>>(out, sink, 1.25); chain_kernel<4><<<1, 64>>>(out, sink, 1.25); chain_kernel<5><<<1, 64>>>(out, sink, 1.25);
    chain_kernel<6><<<1, 64>>>(out, sink, 1.25);
    hipDeviceSynchronize();
  }
  long long h[16];
  hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
  const char* names[] = {"dependent v_fma_f64", "dependent v_rcp_f64 + v_add_f64", "v_readlane pair + v_add_f64", "reciprocal, cubic step (rcp + 3 fma) + add",
                         "reciprocal, two Newton steps (rcp + 4 fma) + add", "pivot link of penta_pipe.h: fma, readlane, reciprocal (cubic)",
                         "pivot link of penta_ldl.h: mul, readlane, fma, readlane, reciprocal (2 Newton)"};
  for (int m = 0; m < 7; ++m) printf("%-80s %lld cycles\n", names[m], h[m]);
  return 0;
}
