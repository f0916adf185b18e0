// microbench: the A* relaxation primitive at the kernel's geometry -- NB one-wave blocks, each walking a frontier of
// 8 nodes x 8 moves over its private 4 MiB g array (row-major 1024 x 1024 u32).  MODE 0: returning atomicMin (what
// astar_kernel does).  MODE 1: L1-bypassing load, then a plain store from the lanes that improve (wave-private data
// needs no atomicity; in-step duplicates would be resolved in LDS).  MODE 2: like 1 with non-returning atomicMin as
// the write.  Reports steps/s and lane-relaxations/s.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int MODE, int TILE> __global__ void __launch_bounds__(64) k(uint32_t* base, int steps, uint32_t* out) {
  uint32_t* g = base + (size_t)blockIdx.x * (1 << 20);
  const int lane = threadIdx.x, d = lane & 7, node = lane >> 3;
  const int dx = (int)((0x2252u >> (2 * d)) & 3u) - 1, dy = (int)((0x0A25u >> (2 * d)) & 3u) - 1;
  uint32_t h = blockIdx.x * 2654435761u + node * 40503u;
  int x = 100 + (h & 511), y = 100 + ((h >> 9) & 511);
  uint32_t acc = 0;
  for (int s = 0; s < steps; ++s) {
    const int nx = (x + dx) & 1023, ny = (y + dy) & 1023;
    uint32_t* p = TILE == 0 ? &g[ny * 1024 + nx]
                : TILE == 1 ? &g[(((ny >> 2) * 128 + (nx >> 3)) << 5) + ((ny & 3) << 3) + (nx & 7)]      // 8 x 4 cells per 128 B line
                            : &g[(((ny >> 2) * 256 + (nx >> 2)) << 4) + ((ny & 3) << 2) + (nx & 3)];     // 4 x 4 cells per 64 B
    const uint32_t nv = 0x40000000u - s * 8u - d;            // decreasing: about every relaxation "improves"
    const bool tryit = ((h >> (d + 3)) & 3u) != 0;           // ~75 % of the moves are legal
    uint32_t old = 0;
    if (MODE == 0) {
      if (tryit) old = __hip_atomic_fetch_min(p, nv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (tryit) old = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const bool imp = tryit && old > nv && ((h >> d) & 1u);  // ~half of them store
      if (MODE == 1) { if (imp) __hip_atomic_store(p, nv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      else { if (imp) __hip_atomic_fetch_min(p, nv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    }
    acc += old;
    // the frontier moves on: next positions depend on this step's results (like the open list)
    h = h * 1664525u + 1013904223u + (acc & 1u);
    const uint32_t hh = __shfl(h, node * 8);
    x = (x + (int)(hh & 3u) - 1) & 1023; y = (y + (int)((hh >> 2) & 3u) - 1) & 1023;
  }
  if (acc == 0x12345u) out[0] = acc;
}
template <int MODE, int TILE> void run(const char* name, uint32_t* d, uint32_t* o, int nb, int steps) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  (void)hipMemset(d, 0xFF, (size_t)nb << 22);
  k<MODE, TILE><<<nb, 64>>>(d, steps / 4, o); (void)hipDeviceSynchronize();
  (void)hipEventRecord(a); k<MODE, TILE><<<nb, 64>>>(d, steps, o); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  printf("%-28s blocks=%4d: %.3f us/step/wave  %.2f G steps/s  %.1f G lane-relaxations/s\n", name, nb, ms * 1e3 / steps,
         (double)nb * steps / (ms * 1e-3) / 1e9, (double)nb * steps * 48 / (ms * 1e-3) / 1e9);
}
int main() {
  const int NBMAX = 4096;
  uint32_t *d, *o; (void)hipMalloc(&d, (size_t)NBMAX << 22); (void)hipMalloc(&o, 4);
  for (int nb : {1024, 2048, 4096}) {
    run<0, 0>("atomicMin returning", d, o, nb, 4000);
    run<0, 1>("atomicMin, 8x4 tiles", d, o, nb, 4000);
    run<0, 2>("atomicMin, 4x4 tiles", d, o, nb, 4000);
    run<1, 0>("load + plain store", d, o, nb, 4000);
    run<1, 1>("load + plain store, 8x4 tiles", d, o, nb, 4000);
  }
  return 0;
}
