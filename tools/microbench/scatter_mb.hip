// microbench: cost of scattered (one cache line per lane) returning atomics / L1-bypassing loads, the primitives of
// the A* kernel, at its geometry: 1024 one-wave blocks (one wave per SIMD), each on a private 4 MiB region.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
// MODE 0: NI independent returning atomicMin per step, then one dependent step (address from result)
// MODE 1: same with sc1 (agent-scope) loads      MODE 2: plain loads
template <int MODE, int NI> __global__ void __launch_bounds__(64) k(uint32_t* base, int steps, int active, uint32_t* out) {
  uint32_t* g = base + (size_t)blockIdx.x * (1 << 20);
  const int lane = threadIdx.x;
  uint32_t idx = (lane * 2654435761u + blockIdx.x * 97u) & ((1 << 20) - 1);
  uint32_t acc = 0;
  if (lane < active)
    for (int s = 0; s < steps; ++s) {
      uint32_t r[NI];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        uint32_t a = (idx + i * 1024u + (i * 37u)) & ((1 << 20) - 1);   // "neighbour rows": different lines
        if (MODE == 0) r[i] = __hip_atomic_fetch_min(&g[a], 0x80000000u + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (MODE == 1) r[i] = __hip_atomic_load(&g[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else r[i] = g[a];
      }
#pragma unroll
      for (int i = 0; i < NI; ++i) acc += r[i];
      idx = (idx + 33u + (acc & 1u)) & ((1 << 20) - 1);   // dependent: next step's addresses need this step's results
    }
  if (acc == 0x12345u) out[0] = acc;
}
template <int MODE, int NI> void run(const char* name, uint32_t* d, uint32_t* o, int steps, int active) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  k<MODE, NI><<<1024, 64>>>(d, steps / 4, active, o); (void)hipDeviceSynchronize();
  (void)hipEventRecord(a); k<MODE, NI><<<1024, 64>>>(d, steps, active, o); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  printf("%-16s NI=%d active=%2d: %.3f us/step  %.1f ns per instr  (%.1f G lane-requests/s chip-wide)\n", name, NI, active,
         ms * 1e3 / steps, ms * 1e6 / steps / NI, 1024.0 * active * NI * steps / (ms * 1e-3) / 1e9);
}
int main() {
  uint32_t *d, *o; (void)hipMalloc(&d, (size_t)1024 << 22); (void)hipMalloc(&o, 4); (void)hipMemset(d, 0xFF, (size_t)1024 << 22);
  const int S = 2000;
  for (int act : {64, 16, 4}) {
    run<0, 8>("atomicMin ret", d, o, S, act); run<0, 3>("atomicMin ret", d, o, S, act); run<0, 1>("atomicMin ret", d, o, S, act);
    run<1, 8>("load sc1", d, o, S, act); run<1, 3>("load sc1", d, o, S, act); run<1, 1>("load sc1", d, o, S, act);
    run<2, 8>("load plain", d, o, S, act); run<2, 1>("load plain", d, o, S, act);
  }
  return 0;
}
