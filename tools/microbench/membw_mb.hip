// microbench: achievable streaming write / read / copy bandwidth for EDT-sized buffers (rocprof-free, hipEvent timed)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void __launch_bounds__(256) wr(int4* p, size_t n) { size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; size_t s = (size_t)gridDim.x * 256; for (; i < n; i += s) p[i] = make_int4((int)i, 1, 2, 3); }
__global__ void __launch_bounds__(256) rd(const int4* p, size_t n, int* o) { size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; size_t s = (size_t)gridDim.x * 256; int a = 0; for (; i < n; i += s) { int4 v = p[i]; a ^= v.x ^ v.y ^ v.z ^ v.w; } if (a == 0x12345) *o = a; }
__global__ void __launch_bounds__(256) cp(const int4* p, int4* q, size_t n) { size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; size_t s = (size_t)gridDim.x * 256; for (; i < n; i += s) q[i] = p[i]; }
// read 1 B / write 4 B per element, like EDT: u8 in -> i32 out
__global__ void __launch_bounds__(256) r1w4(const uint32_t* p, int4* q, size_t n4) { size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; size_t s = (size_t)gridDim.x * 256; for (; i < n4; i += s) { uint32_t v = p[i]; q[i] = make_int4(v & 255, (v >> 8) & 255, (v >> 16) & 255, v >> 24); } }
template <class F> float timeit(F f, int it = 20) { hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b); f(); f(); (void)hipDeviceSynchronize(); (void)hipEventRecord(a); for (int i = 0; i < it; ++i) f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b); float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / it; }
int main() {
  size_t MB = 1 << 20; char *A, *B; int* o; (void)hipMalloc(&A, 512 * MB); (void)hipMalloc(&B, 512 * MB); (void)hipMalloc(&o, 4); (void)hipMemset(A, 1, 512 * MB);
  for (int blocks : {2048, 4096, 16384}) {
    for (size_t sz : {64 * MB, 256 * MB}) {
      size_t n = sz / 16;
      float w = timeit([&] { wr<<<blocks, 256>>>((int4*)B, n); });
      float r = timeit([&] { rd<<<blocks, 256>>>((int4*)A, n, o); });
      float c = timeit([&] { cp<<<blocks, 256>>>((int4*)A, (int4*)B, n); });
      printf("blocks=%5d size=%3zu MiB: write %.1f us (%.2f TB/s)  read %.1f us (%.2f TB/s)  copy %.1f us (%.2f TB/s r+w)\n", blocks, sz / MB, w * 1e3, sz / (w * 1e-3) / 1e12, r * 1e3, sz / (r * 1e-3) / 1e12, c * 1e3, 2 * sz / (c * 1e-3) / 1e12);
    }
    size_t n4 = 64 * MB / 4;
    float e = timeit([&] { r1w4<<<blocks, 256>>>((uint32_t*)A, (int4*)B, n4); });
    printf("blocks=%5d EDT-shaped 64 MiB in -> 256 MiB out: %.1f us (%.2f TB/s)\n", blocks, e * 1e3, 320 * MB / (e * 1e-3) / 1e12);
  }
  return 0;
}
