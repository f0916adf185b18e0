// microbench: LDS cost of the EDT kernels' access patterns, 16 wavefronts per CU (1024-thread blocks, one per CU)
//   0: ds_write_b16, lane-contiguous (vertical pass: 2 distance bytes per thread and row)
//   1: ds_write_b32, lane-contiguous
//   2: the packed transposition: 2 x ds_write_b128 + 4 x ds_read_b128 per tile (swizzled layout of edt_band_g8_kernel)
//   3: ds_read_b128, lane-contiguous (a row of distance bytes into registers)
//   4: like 2 but ds_write_b64 x 4 instead of ds_write_b128 x 2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u2_t __attribute__((ext_vector_type(2)));
template <int MODE> __global__ void __launch_bounds__(1024) k(uint32_t* out, int n) {
    extern __shared__ uint32_t sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t* tr = sm + wave * 1024;   // 4 KiB per wave
    uint32_t acc = threadIdx.x;
    u4_t a = {acc, acc + 1, acc + 2, acc + 3}, b = {acc + 4, acc + 5, acc + 6, acc + 7};
    for (int it = 0; it < n; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) reinterpret_cast<volatile uint16_t*>(tr)[r * 128 + lane] = (uint16_t)(acc + r);
        } else if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) reinterpret_cast<volatile uint32_t*>(tr)[r * 64 + lane] = acc + r;
        } else if (MODE == 2 || MODE == 4) {
            uint32_t* wb = tr + 8 * lane;
            const int s = (lane >> 2) & 1;
            if (MODE == 2) {
                *reinterpret_cast<volatile u4_t*>(wb + 4 * s) = a;
                *reinterpret_cast<volatile u4_t*>(wb + 4 * (1 - s)) = b;
            } else {
                *reinterpret_cast<volatile u2_t*>(wb + 4 * s) = u2_t{a.x, a.y};
                *reinterpret_cast<volatile u2_t*>(wb + 4 * s + 2) = u2_t{a.z, a.w};
                *reinterpret_cast<volatile u2_t*>(wb + 4 * (1 - s)) = u2_t{b.x, b.y};
                *reinterpret_cast<volatile u2_t*>(wb + 4 * (1 - s) + 2) = u2_t{b.z, b.w};
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int l2 = 16 * kk + lane / 4;
                const u4_t q = *reinterpret_cast<volatile u4_t*>(tr + 8 * l2 + 4 * ((lane & 1) ^ ((lane >> 4) & 1)));
                a.x += q.x; a.y += q.y; b.z += q.z; b.w += q.w;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (MODE == 3) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const u4_t q = *reinterpret_cast<volatile u4_t*>(tr + 256 * r + 4 * lane);
                a.x += q.x; a.y += q.y; b.z += q.z; b.w += q.w;
            }
        }
    }
    out[blockIdx.x * 1024 + threadIdx.x] = a.x + a.y + b.z + b.w + acc;
}
template <int MODE> void run(const char* name, uint32_t* d, int insts_per_iter) {
    const int n = 2000, blocks = 256;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 64 * 1024, 0, d, n); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 64 * 1024, 0, d, n); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // per CU: 16 waves x n x insts_per_iter LDS instructions in ms
    printf("%-44s %.3f ms  -> %.1f cycles per LDS wave-instruction per CU (@2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / (16.0 * n * insts_per_iter));
}
int main() {
    uint32_t* d; (void)hipMalloc(&d, 256 * 1024 * 4);
    run<0>("ds_write_b16 lane-contiguous", d, 16);
    run<1>("ds_write_b32 lane-contiguous", d, 16);
    run<2>("transposition 2 x write_b128 + 4 x read_b128", d, 6);
    run<4>("transposition 4 x write_b64 + 4 x read_b128", d, 8);
    run<3>("ds_read_b128 lane-contiguous", d, 4);
    return 0;
}
