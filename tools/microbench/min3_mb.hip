// microbench: (1) is v_pk_minimum3_f16 an exact unsigned min3 on 16-bit patterns 0 .. 0x7BFF (positive f16, denormals
// included)?  (2) cost of one cascade iteration in the 3-op form (min, add_sat, min) and in the 2-op form (add, min3).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pkmin(uint32_t a, uint32_t b){ return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(us2,a), __builtin_bit_cast(us2,b))); }
__device__ __forceinline__ uint32_t pkadds(uint32_t a, uint32_t b){ return __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(us2,a), __builtin_bit_cast(us2,b))); }
__device__ __forceinline__ uint32_t pkadd(uint32_t a, uint32_t b){ return __builtin_bit_cast(uint32_t, __builtin_bit_cast(us2,a) + __builtin_bit_cast(us2,b)); }
__device__ __forceinline__ uint32_t pkmin3(uint32_t a, uint32_t b, uint32_t c){ uint32_t r; asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

__global__ void check_kernel(unsigned long long* bad, uint32_t* first_bad) {
    // a = every value 0 .. 0x7BFF (low half) paired with a hashed value (high half); b, c hashed; plus the edge cases
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t h = tid * 2654435761u + 12345u;
    unsigned long long nb = 0;
    for (int r = 0; r < 256; ++r) {
        h = h * 1664525u + 1013904223u; const uint32_t a0 = tid % 0x7C00u, a1 = (h >> 8) % 0x7C00u;
        h = h * 1664525u + 1013904223u; uint32_t b0 = (h >> 8) % 0x7C00u, b1 = (h >> 4) % 0x7C00u;
        h = h * 1664525u + 1013904223u; uint32_t c0 = (h >> 8) % 0x7C00u, c1 = (h >> 4) % 0x7C00u;
        if (r < 64) { b0 = (a0 + r) % 0x7C00u; c0 = (a0 + 0x7C00u - r) % 0x7C00u; }   // near-equal values
        if (r == 64) { b0 = 0; c0 = 0x7BFF; } if (r == 65) { b0 = 0x3FF; c0 = 0x400; } if (r == 66) { b0 = 1; c0 = 2; }
        const uint32_t a = a0 | (a1 << 16), b = b0 | (b1 << 16), c = c0 | (c1 << 16);
        const uint32_t got = pkmin3(a, b, c), want = pkmin(a, pkmin(b, c));
        if (got != want) { if (!nb && atomicCAS(first_bad, 0u, 1u) == 0u) { first_bad[1] = a; first_bad[2] = b; first_bad[3] = c; first_bad[4] = got; first_bad[5] = want; } ++nb; }
    }
    if (nb) atomicAdd(bad, nb);
}

template<int MODE> __global__ void __launch_bounds__(512, 8) k(uint32_t* p, int n){
  uint32_t P[8];
  for(int j=0;j<8;++j) P[j]=p[(blockIdx.x*512+threadIdx.x)*8+j] & 0x0FFF0FFFu;
  for(int it=1; it<=n; ++it){
    const uint32_t c=(uint32_t)(2*it-1)*0x00010001u & 0x01FF01FFu;
    if (MODE == 0) {
      const uint32_t T=P[7], S=P[0];
      const uint32_t below=__builtin_amdgcn_update_dpp(-1, (int)T, 0x138, 0xF, 0xF, false), above=__builtin_amdgcn_update_dpp(-1,(int)S, 0x130, 0xF, 0xF, false);
      const uint32_t L0=__builtin_amdgcn_alignbit(T, below, 16), RL=__builtin_amdgcn_alignbit(above, S, 16);
      uint32_t m[8];
#pragma unroll
      for(int j=0;j<8;++j) m[j]=pkmin(j?P[j-1]:L0, j<7?P[j+1]:RL);
#pragma unroll
      for(int j=0;j<8;++j) m[j]=pkadds(m[j],c);
#pragma unroll
      for(int j=0;j<8;++j) P[j]=pkmin(P[j],m[j]);
    } else {
      uint32_t T[8];
#pragma unroll
      for(int j=0;j<8;++j) T[j]=pkadd(P[j],c);
      const uint32_t below=__builtin_amdgcn_update_dpp(0x7BFF7BFF, (int)T[7], 0x138, 0xF, 0xF, false), above=__builtin_amdgcn_update_dpp(0x7BFF7BFF,(int)T[0], 0x130, 0xF, 0xF, false);
      const uint32_t L0=__builtin_amdgcn_alignbit(T[7], below, 16), RL=__builtin_amdgcn_alignbit(above, T[0], 16);
#pragma unroll
      for(int j=0;j<8;++j) P[j]=pkmin3(P[j], j?T[j-1]:L0, j<7?T[j+1]:RL);
    }
  }
  for(int j=0;j<8;++j) p[(blockIdx.x*512+threadIdx.x)*8+j]=P[j];
}
template<int MODE> void run(const char* name, uint32_t* d, int blocks, int n){
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<blocks,512>>>(d,n); hipDeviceSynchronize();
  hipEventRecord(a); k<MODE><<<blocks,512>>>(d,n); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms,a,b);
  double waves=blocks*8.0; double per_simd_waves=waves/1024.0;
  printf("%-28s blocks=%d n=%d  %.3f ms  -> %.1f SIMD-cycles per wave-iteration (@2.4GHz)\n", name, blocks, n, ms, ms*1e-3*2.4e9/(n*per_simd_waves));
}
int main(){
  unsigned long long* bad; uint32_t* fb; hipMalloc(&bad, 8); hipMalloc(&fb, 32); hipMemset(bad, 0, 8); hipMemset(fb, 0, 32);
  check_kernel<<<0x7C00 * 4 / 256, 256>>>(bad, fb); hipDeviceSynchronize();
  unsigned long long hb; uint32_t hf[8]; hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(hf, fb, 32, hipMemcpyDeviceToHost);
  printf("v_pk_minimum3_f16 as u16 min3 on 0..0x7BFF: %llu mismatches of %llu", hb, (unsigned long long)0x7C00 * 4 * 256);
  if (hb) printf("  first: a=%08x b=%08x c=%08x got=%08x want=%08x", hf[1], hf[2], hf[3], hf[4], hf[5]);
  printf("\n");
  uint32_t* d; int blocks=256*4; hipMalloc(&d, (size_t)blocks*512*8*4); hipMemset(d, 0x11, (size_t)blocks*512*8*4);
  int n=4000;
  run<0>("3-op (min, add_sat, min)", d, blocks, n);
  run<1>("2-op (add, min3_f16)", d, blocks, n);
  run<0>("3-op (min, add_sat, min)", d, blocks, n);
  run<1>("2-op (add, min3_f16)", d, blocks, n);
  return 0; }
