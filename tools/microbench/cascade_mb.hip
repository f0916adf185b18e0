// microbench: cost of the cascade iteration variants on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pkmin(uint32_t a, uint32_t b){ return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(us2,a), __builtin_bit_cast(us2,b))); }
__device__ __forceinline__ uint32_t pkadds(uint32_t a, uint32_t b){ return __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(us2,a), __builtin_bit_cast(us2,b))); }
template<int MODE> __global__ void __launch_bounds__(256) k(uint32_t* p, int n){
  uint32_t P[8];
  for(int j=0;j<8;++j) P[j]=p[(blockIdx.x*256+threadIdx.x)*8+j];
  for(int it=1; it<=n; ++it){
    uint32_t c=(2*it-1); if(MODE!=2) c|=c<<16;
    uint32_t T=P[7], S=P[0], below, above;
    if(MODE==0||MODE==2){ below=__builtin_amdgcn_update_dpp(-1, (int)T, 0x138, 0xF, 0xF, false); above=__builtin_amdgcn_update_dpp(-1,(int)S, 0x130, 0xF, 0xF, false);} 
    else if(MODE==3){ below=__builtin_amdgcn_update_dpp(-1, (int)T, 0x111, 0xF, 0xF, false); above=__builtin_amdgcn_update_dpp(-1,(int)S, 0x101, 0xF, 0xF, false);} // row_shr:1,row_shl:1
    else if(MODE==4){ below=__shfl_up(T,1); above=__shfl_down(S,1);} 
    else { below=T^it; above=S^it; }
    uint32_t L0=__builtin_amdgcn_alignbit(T, below, 16), RL=__builtin_amdgcn_alignbit(above, S, 16);
    uint32_t prev=L0;
#pragma unroll
    for(int j=0;j<8;++j){ uint32_t cur=P[j]; uint32_t nxt=j<7?P[j+1]:RL;
      if(MODE==2) P[j]=min(cur, min(prev,nxt)+c); else P[j]=pkmin(cur, pkadds(pkmin(prev,nxt), c)); prev=cur; }
  }
  for(int j=0;j<8;++j) p[(blockIdx.x*256+threadIdx.x)*8+j]=P[j];
}
template<int MODE> void run(const char* name, uint32_t* d, int blocks, int n){
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<blocks,256>>>(d,n); hipDeviceSynchronize();
  hipEventRecord(a); k<MODE><<<blocks,256>>>(d,n); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms,a,b);
  double waves=blocks*4.0; double per_simd_waves=waves/1024.0; // 256 CUs * 4 SIMDs
  double cyc=ms*1e-3*2.4e9/(n*per_simd_waves);
  printf("%-28s blocks=%d n=%d  %.3f ms  -> %.1f SIMD-cycles per wave-iteration (@2.4GHz)\n", name, blocks, n, ms, cyc);
}
int main(){ uint32_t* d; int blocks=256*8; hipMalloc(&d, (size_t)blocks*256*8*4); hipMemset(d, 0x11, (size_t)blocks*256*8*4);
  int n=4000;
  run<0>("packed + wave_shr dpp", d, blocks, n);
  run<1>("packed, no dpp", d, blocks, n);
  run<2>("u32 + wave_shr dpp", d, blocks, n);
  run<3>("packed + row_shr dpp", d, blocks, n);
  run<4>("packed + shfl (bpermute)", d, blocks, n);
  run<0>("packed + wave_shr, 1 blk/CU", d, 256, n);
  return 0; }
