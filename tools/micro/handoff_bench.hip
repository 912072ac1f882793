// Microbenchmarks behind DESIGN.md 4.3 / 11 (what a launch boundary, an in-kernel hand-over and a few
// instruction patterns cost on this machine).  hipcc --offload-arch=gfx950 -O3 handoff_bench.hip -o handoff_bench
//   1. N dependent launches of an empty kernel on one stream: microseconds per launch.
//   2. producer workgroup -> consumer workgroup on another CU, 4 KB payload + flag, (a) plain stores + release
//      fence + relaxed flag, (b) agent-scope (sc1, write-through) stores + s_waitcnt vmcnt(0) + flag; the
//      consumer polls with relaxed loads, acquires, reads the payload: microseconds from the producer's first
//      store to the consumer's last load (wall_clock64 on both sides, 100 MHz).
//   3. one wavefront: ds_read_b128 aligned vs 8 bytes off; a chain of dependent v_add_f64 / v_fma_f64;
//      64 lanes storing 8 bytes each with stride 8 B (coalesced) vs 24 B (scattered); cycles per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 12345) *p = 1; }

__global__ void __launch_bounds__(256) handoff_kernel(double* payload, unsigned* flag, long long* stamps, int mode, unsigned epoch) {
  const int tid = threadIdx.x;
  if (blockIdx.x == 0) {   // producer
    __builtin_amdgcn_s_sleep(100);
    if (tid == 0) stamps[0] = wall_clock64();
    const double v = (double)epoch + tid;
    if (mode == 0) {
      payload[tid] = v; payload[256 + tid] = v;
      __syncthreads();
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      __hip_atomic_store(payload + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(payload + 256 + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else if (blockIdx.x == gridDim.x - 1) {   // consumer (last workgroup: another CU, most likely another XCD)
    if (tid == 0)
      while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) __builtin_amdgcn_s_sleep(1);
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const double a = payload[tid] + payload[256 + tid];
    if (a != 2.0 * ((double)epoch + tid)) stamps[3] = -1;   // stale data would show here
    __syncthreads();
    if (tid == 0) stamps[1] = wall_clock64();
  }
}

__global__ void __launch_bounds__(64) instr_kernel(long long* out, double* gbuf) {
  __shared__ double lds[4096];
  const int lane = threadIdx.x;
  for (int i = lane; i < 4096; i += 64) lds[i] = i;
  __syncthreads();
  typedef double __attribute__((ext_vector_type(2))) v2d;
  v2d acc = {0, 0};
  long long t0 = clock64();
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    v2d t;
    asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(t) : "v"((unsigned)((lane * 2 + i * 128) * 8)));            // aligned
    acc += t;
  }
  long long t1 = clock64();
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    v2d t;
    asm volatile("ds_read_b128 %0, %1 offset:8\n s_waitcnt lgkmcnt(0)" : "=v"(t) : "v"((unsigned)((lane * 2 + i * 128) * 8)));   // 8 bytes off
    acc += t;
  }
  long long t2 = clock64();
  double x = acc.x + 1.0, c1 = 1.25 + acc.y * 1e-300;
#pragma unroll
  for (int i = 0; i < 64; ++i) { x = x + c1; asm volatile("" : "+v"(x)); }          // dependent v_add_f64
  long long t3 = clock64();
#pragma unroll
  for (int i = 0; i < 64; ++i) { x = __builtin_fma(x, c1, 0.5); asm volatile("" : "+v"(x)); }   // dependent v_fma_f64
  long long t4 = clock64();
  const long long t4a = t4;
#pragma unroll
  for (int i = 0; i < 64; ++i) { x = x * c1; asm volatile("" : "+v"(x)); }           // dependent v_mul_f64
  long long t4b = clock64();
  double y0 = x, y1 = x + 1, y2 = x + 2, y3 = x + 3, y4 = x + 4, y5 = x + 5, y6 = x + 6, y7 = x + 7;
  long long t4c = clock64();
#pragma unroll
  for (int i = 0; i < 16; ++i) {                                                       // 8 independent v_fma_f64 chains
    y0 = __builtin_fma(y0, c1, 0.5); y1 = __builtin_fma(y1, c1, 0.5); y2 = __builtin_fma(y2, c1, 0.5); y3 = __builtin_fma(y3, c1, 0.5);
    y4 = __builtin_fma(y4, c1, 0.5); y5 = __builtin_fma(y5, c1, 0.5); y6 = __builtin_fma(y6, c1, 0.5); y7 = __builtin_fma(y7, c1, 0.5);
    asm volatile("" : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7));
  }
  long long t4d = clock64();
#pragma unroll
  for (int i = 0; i < 16; ++i) {                                                       // 8 independent v_add_f64 chains
    y0 = y0 + c1; y1 = y1 + c1; y2 = y2 + c1; y3 = y3 + c1; y4 = y4 + c1; y5 = y5 + c1; y6 = y6 + c1; y7 = y7 + c1;
    asm volatile("" : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7));
  }
  long long t4f = clock64();
#pragma unroll
  for (int i = 0; i < 16; ++i) {                                                       // 8 independent v_mul_f64 chains
    y0 = y0 * c1; y1 = y1 * c1; y2 = y2 * c1; y3 = y3 * c1; y4 = y4 * c1; y5 = y5 * c1; y6 = y6 * c1; y7 = y7 * c1;
    asm volatile("" : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7));
  }
  long long t4g = clock64();
  x = ((y0 + y1) + (y2 + y3)) + ((y4 + y5) + (y6 + y7));
  long long t4e = clock64();
  (void)t4e;
  t4 = clock64();
#pragma unroll
  for (int i = 0; i < 16; ++i) gbuf[i * 4096 + lane] = x;             // coalesced 8-byte stores
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  long long t5 = clock64();
#pragma unroll
  for (int i = 0; i < 16; ++i) gbuf[65536 + i * 4096 + lane * 3] = x;  // stride 24 B
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  long long t6 = clock64();
  if (lane == 0) {
    out[0] = (t1 - t0) / 32; out[1] = (t2 - t1) / 32; out[2] = (t3 - t2) / 64; out[3] = (t4a - t3) / 64; out[8] = (t4b - t4a) / 64;
    out[4] = (t5 - t4) / 16; out[5] = (t6 - t5) / 16; out[6] = (t4d - t4c) / 128; out[9] = (t4f - t4d) / 128; out[10] = (t4g - t4f) / 128;
  }
  if (x == 12345.6789 && acc.y == 1.0) out[7] = 1;
}

int main() {
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, (int*)nullptr);
  hipStreamSynchronize(s);
  for (int grid : {1, 40, 164}) {
    hipEventRecord(a, s);
    for (int i = 0; i < 500; ++i) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(256), 0, s, (int*)nullptr);
    hipEventRecord(b, s); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::printf("500 dependent launches of an empty kernel, grid %3d x 256: %.2f us per launch\n", grid, 1e3 * ms / 500);
  }
  double* payload; unsigned* flag; long long* stamps;
  hipMalloc(&payload, 512 * 8); hipMalloc(&flag, 64); hipMalloc(&stamps, 64);
  hipMemset(flag, 0, 64); hipMemset(stamps, 0, 64); hipDeviceSynchronize();
  for (int mode = 0; mode < 2; ++mode) {
    double sum = 0, mx = 0; int stale = 0;
    for (unsigned e = 1; e <= 200; ++e) {
      hipLaunchKernelGGL(handoff_kernel, dim3(64), dim3(256), 0, s, payload, flag, stamps, mode, e + 1000 * mode);
      hipStreamSynchronize(s);
      long long h[4]; hipMemcpy(h, stamps, 32, hipMemcpyDeviceToHost);
      const double us = (h[1] - h[0]) * 0.01;
      if (e > 20) { sum += us; if (us > mx) mx = us; }
      if (h[3] != 0) ++stale;
    }
    std::printf("hand-over of 4 KB + flag between two workgroups, %s: %.2f us mean, %.2f max%s\n",
                mode == 0 ? "plain stores + release fence" : "agent-scope (sc1) stores + vmcnt(0)", sum / 180, mx,
                stale ? "  (STALE DATA SEEN)" : "");
  }
  long long* out; double* gbuf; hipMalloc(&out, 128); hipMalloc(&gbuf, 8 * (65536 * 2 + 4096 * 16 * 3)); hipMemset(out, 0, 128);
  hipLaunchKernelGGL(instr_kernel, dim3(1), dim3(64), 0, s, out, gbuf);
  hipStreamSynchronize(s);
  long long h[16]; hipMemcpy(h, out, 128, hipMemcpyDeviceToHost);
  std::printf("one wavefront, cycles per instruction: ds_read_b128 (waited for, each) aligned %lld, 8 bytes off %lld; dependent v_add_f64 %lld, v_fma_f64 %lld, v_mul_f64 %lld, v_fma_f64 / v_add_f64 / v_mul_f64 with 8 independent chains %lld / %lld / %lld; "
              "8-byte global store (16 issued, then vmcnt(0)) coalesced %lld, stride 24 B %lld\n", h[0], h[1], h[2], h[3], h[8], h[6], h[9], h[10], h[4], h[5]);
  return 0;
}
