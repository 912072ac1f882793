// The eliminating wavefront's row of penta_pipe.h (pipe_pivot<19>: K pivots, each published to an LDS ring slot, the
// multipliers read back as LDS broadcasts) ALONE on a compute unit: cycles per row and per pivot, with 0 .. 7 other
// wavefronts of the workgroup that (a) sleep, (b) stream LDS reads of the published rows as the followers do, (c) issue
// dependent f64 FMAs.  Separates what the wavefront costs by itself from what its neighbours cost it.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I../../idto_amd/csrc -I../../include pivot_bench.hip -o pivot_bench
// (-DPIPE_MULT_READLANE: the multipliers by v_readlane instead of LDS broadcasts - 27 % faster HERE, 3.6 % slower in the
// kernel, where the wavefront also carries the follower hand-over and runs out of SGPRs)
#include <hip/hip_runtime.h>
#include <cstdio>
#include "kernels.h"
#include "penta_ldl.h"
#include "penta_nd.h"
#include "penta_pipe.h"

using namespace idto_dev;
constexpr int K = 19;
using G = PipeGeo<K>;

// mode of the other wavefronts: 0 absent (64 threads), 1 sleep, 2 LDS broadcast reads, 3 dependent FMAs
template <int OTHERS>
__global__ void __launch_bounds__(512) pivot_kernel(long long* out, double* sink, int rows, int nothers, double seed) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double* ring = lds;   // one slot
  for (int i = tid; i < G::SLOT; i += blockDim.x) ring[i] = 0.0;
  __shared__ int stop;
  if (tid == 0) stop = 0;
  __syncthreads();
  if (wave == 0) {
    int pos, rawpos = G::odump + (lane & 7);
    if (lane < K) pos = G::oS + lane;
    else if (lane < 2 * K) { pos = G::oH + lane - K; rawpos = G::oRH + lane - K; }
    else if (lane < 3 * K) { pos = G::oE + lane - 2 * K; rawpos = G::oRE + lane - 2 * K; }
    else if (lane == 3 * K) { pos = G::oy; rawpos = G::oRy; }
    else pos = (lane == 62) ? G::od : (lane == 63) ? G::oi : G::odump + (lane & 7);
    double acc = 0.0;
    long long t0 = 0, t1 = 0;
    double x0[K];
#pragma unroll
    for (int r = 0; r < K; ++r) x0[r] = (r == lane ? 4.0 + seed : 0.01 * seed) + 1e-3 * ((r * 7 + lane * 3) % 11);   // diagonally dominant
    for (int it = 0; it < rows + 2; ++it) {
      if (it == 2) t0 = __builtin_readcyclecounter();
      double xr[K];
#pragma unroll
      for (int r = 0; r < K; ++r) xr[r] = x0[r] + acc * 1e-300;   // (a row's inputs depend on the row before, as in the chain)
      double mu0[K];
#pragma unroll
      for (int r = 0; r < K; ++r) mu0[r] = 0.0;
      auto hook = [&]() {};
      pipe_pivot<K, 0, 4>(xr, pipe_rcp(rdlane(xr[0], 0)), mu0, 0.0, ring + pos, ring + rawpos, ring, lane == 63, hook);
      acc += xr[K - 1];
    }
    t1 = __builtin_readcyclecounter();
    if (lane == 0) { out[0] = t1 - t0; stop = 1; }
    sink[tid] = acc;
  } else if (OTHERS != 0 && wave <= nothers) {
    double acc = seed;
    volatile int* st = &stop;
    if (OTHERS == 1) { while (!*st) __builtin_amdgcn_s_sleep(8); }
    if (OTHERS == 2) {   // the followers' traffic: K / 2 broadcast b128 reads + K per-lane reads of a published row per pivot
      const double2* p2 = reinterpret_cast<const double2*>(ring);
      while (!*st) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
          double2 a = p2[(j * G::RS) / 2 + 0], b = p2[(j * G::RS) / 2 + 3], c2 = p2[(j * G::RS) / 2 + 6];
          acc += a.x + b.y + c2.x + ring[j * G::RS + G::oRH + (lane % K)];
        }
      }
    }
    if (OTHERS == 3) {
      double b = 1.0000001;
      while (!*st) {
#pragma unroll
        for (int j = 0; j < 64; ++j) acc = __builtin_fma(acc, b, 1e-9);
      }
    }
    sink[tid] = acc;
  }
}

int main() {
  long long* out; double* sink;
  hipMalloc(&out, 64); hipMalloc(&sink, 512 * 8);
  const int rows = 64, lds = G::SLOT * 8 + 64;
  auto run = [&](auto kern, int threads, int nothers, const char* what) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    long long best = 1ll << 60;
    for (int rep = 0; rep < 5; ++rep) {
      hipLaunchKernelGGL(kern, dim3(1), dim3(threads), lds, 0, out, sink, rows, nothers, 1.0 + rep * 1e-6);
      long long c = 0;
      hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost);
      if (c < best) best = c;
    }
    printf("%-78s %7.0f cycles per row of %d pivots, %6.1f per pivot\n", what, (double)best / rows, K, (double)best / rows / K);
  };
  run(pivot_kernel<0>, 64, 0, "pipe_pivot<19> alone in its workgroup");
  run(pivot_kernel<1>, 512, 7, "... with 7 sleeping wavefronts");
  run(pivot_kernel<2>, 128, 1, "... with 1 wavefront streaming LDS reads (another SIMD)");
  run(pivot_kernel<2>, 256, 3, "... with 3 wavefronts streaming LDS reads (the other three SIMDs)");
  run(pivot_kernel<2>, 512, 7, "... with 7 wavefronts streaming LDS reads (one of them on its SIMD)");
  run(pivot_kernel<3>, 256, 3, "... with 3 wavefronts of dependent f64 FMAs (the other three SIMDs)");
  run(pivot_kernel<3>, 512, 7, "... with 7 wavefronts of dependent f64 FMAs (one of them on its SIMD)");
  return 0;
}
