// Where do the workgroups of a launch land, and what does a hand-over between two of them cost when they share an XCD
// (one L2) against when they do not?  hipcc --offload-arch=gfx950 -O3 xcd_pingpong.hip -o xcd_pingpong
//   1. XCC_ID (s_getreg) of workgroup i of a 1-D grid: is it i mod 8?
//   2. ping-pong of one word between workgroups a and b (the others exit at once), R round trips, microseconds per
//      round trip (wall_clock64, 100 MHz):  mode 0: agent-scope atomics (sc1: through the L2 to the fabric),
//      mode 1: workgroup-scope atomics (sc0: L1 bypassed, served by the XCD's L2 - only meaningful on ONE XCD).
//   3. mode 1 with a payload: 64 doubles written with plain stores, s_waitcnt vmcnt(0), then the flag; the reader
//      polls the flag (sc0) and reads the payload with sc0 loads; mismatches are counted.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void xcc_kernel(int* out) {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  if (threadIdx.x == 0) out[blockIdx.x] = (int)(v & 0xf);
}

template <int SCOPE>
__device__ __forceinline__ unsigned ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, SCOPE); }
template <int SCOPE>
__device__ __forceinline__ void st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, SCOPE); }

template <int SCOPE>
__device__ bool wait_for(const unsigned* p, unsigned v) {
  for (int i = 0; i < 2000000; ++i) if (ld<SCOPE>(p) == v) return true;
  return false;
}

template <int SCOPE, bool PAYLOAD>
__global__ void __launch_bounds__(64) pingpong_kernel(unsigned* w, double* payload, long long* out, int a, int b, int R, unsigned base) {
  const int me = blockIdx.x, lane = threadIdx.x;
  if (me != a && me != b) return;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  long long bad = 0, lost = 0;
  const long long t0 = wall_clock64();
  for (int i = 1; i <= R; ++i) {
    const unsigned v = base + i;
    if (me == a) {
      if (PAYLOAD) { payload[lane] = (double)v + lane; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      if (lane == 0) st<SCOPE>(w, v);
      if (!wait_for<SCOPE>(w + 32, v)) { ++lost; break; }
    } else {
      if (!wait_for<SCOPE>(w, v)) { ++lost; break; }
      if (PAYLOAD) {
        const double got = __hip_atomic_load(payload + lane, __ATOMIC_RELAXED, SCOPE);
        if (got != (double)v + lane) ++bad;
      }
      if (lane == 0) st<SCOPE>(w + 32, v);
    }
  }
  const long long t1 = wall_clock64();
  if (lane == 0) {
    const int o = (me == a) ? 0 : 4;
    out[o] = t1 - t0; out[o + 1] = (long long)(xcc & 0xf); out[o + 2] = lost;
  }
  if (PAYLOAD && me == b) atomicAdd((unsigned long long*)(out + 3), (unsigned long long)bad);
}

int main() {
  int* xo; hipMalloc(&xo, 64 * sizeof(int));
  xcc_kernel<<<64, 64>>>(xo);
  std::vector<int> h(64);
  hipMemcpy(h.data(), xo, 64 * sizeof(int), hipMemcpyDeviceToHost);
  printf("XCC_ID of workgroups 0..63:");
  bool rr = true;
  for (int i = 0; i < 64; ++i) { printf(" %d", h[i]); rr = rr && h[i] == i % 8; }
  printf("\n  round robin (i mod 8): %s\n", rr ? "yes" : "NO");
  // a second launch right after (does the round robin restart at XCD 0 for every dispatch?)
  xcc_kernel<<<13, 64>>>(xo); xcc_kernel<<<64, 64>>>(xo);
  hipMemcpy(h.data(), xo, 64 * sizeof(int), hipMemcpyDeviceToHost);
  rr = true;
  for (int i = 0; i < 64; ++i) rr = rr && h[i] == i % 8;
  printf("  after a 13-workgroup launch, the next launch starts at XCD 0 again: %s (wg0 on %d)\n", rr ? "yes" : "NO", h[0]);

  unsigned* w; double* payload; long long* out;
  hipMalloc(&w, 4096); hipMalloc(&payload, 4096); hipMalloc(&out, 64);
  hipMemset(w, 0, 4096); hipMemset(payload, 0, 4096);
  const int R = 2000;
  unsigned base = 0;
  auto run = [&](const char* what, int mode, int a, int b) {
    hipMemset(out, 0, 64);
    if (mode == 0) pingpong_kernel<__HIP_MEMORY_SCOPE_AGENT, false><<<64, 64>>>(w, payload, out, a, b, R, base);
    if (mode == 1) pingpong_kernel<__HIP_MEMORY_SCOPE_WORKGROUP, false><<<64, 64>>>(w, payload, out, a, b, R, base);
    if (mode == 2) pingpong_kernel<__HIP_MEMORY_SCOPE_AGENT, true><<<64, 64>>>(w, payload, out, a, b, R, base);
    if (mode == 3) pingpong_kernel<__HIP_MEMORY_SCOPE_WORKGROUP, true><<<64, 64>>>(w, payload, out, a, b, R, base);
    base += R + 1;
    long long o[8];
    hipDeviceSynchronize();
    hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost);
    printf("%-58s wg %2d (xcd %lld) <-> wg %2d (xcd %lld): %.3f us per round trip%s%s", what, a, o[1], b, o[5], o[0] * 0.01 / R,
           (o[2] || o[6]) ? "  WAIT RAN OUT (stale)" : "", "");
    if (mode >= 2) printf("  payload mismatches %lld", o[3]);
    printf("\n");
  };
  for (int rep = 0; rep < 2; ++rep) {
    run("agent-scope word, different XCDs", 0, 0, 1);
    run("agent-scope word, same XCD", 0, 0, 8);
    run("workgroup-scope (sc0) word, same XCD", 1, 0, 8);
    run("workgroup-scope (sc0) word, different XCDs (expected to fail)", 1, 0, 1);
    run("agent-scope word + 512 B payload (sc1 loads), different XCDs", 2, 0, 1);
    run("agent-scope word + 512 B payload (sc1 loads), same XCD", 2, 0, 8);
    run("sc0 word + 512 B payload (plain stores, sc0 loads), same XCD", 3, 0, 8);
  }
  return 0;
}
