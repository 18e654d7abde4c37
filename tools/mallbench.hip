// Development micro-benchmark (not part of libhadi): ping-pong streaming copy a->b, b->a as a function of the working
// set, with the second copy walking ascending or descending -- what does the 256 MB memory-side cache give?
//   hipcc -O3 --offload-arch=gfx950 tools/mallbench.hip -o tools/mallbench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2v __attribute__((ext_vector_type(2)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)
// each block copies one contiguous 64 KiB piece; pieces are taken ascending or descending by blockIdx
template<int NT> __global__ void __launch_bounds__(256) copyk(const d2v* __restrict__ in, d2v* __restrict__ out, size_t npieces, int desc){
  size_t p = desc ? npieces-1-blockIdx.x : blockIdx.x;
  const d2v* s=in+p*4096; d2v* d=out+p*4096;
  #pragma unroll 4
  for(int i=threadIdx.x;i<4096;i+=256){ d2v t = NT ? __builtin_nontemporal_load(s+i) : s[i]; t.x*=1.0000001; if(NT>1) __builtin_nontemporal_store(t,d+i); else d[i]=t; }
}
int main(){
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms; const int reps=20;
  for(size_t mb : {16,32,64,96,128,192,256,384,512,1024}){
    size_t bytes=mb*1024*1024/2; size_t npieces=bytes/65536; double *a,*b; CK(hipMalloc(&a,bytes)); CK(hipMalloc(&b,bytes)); CK(hipMemset(a,0,bytes)); CK(hipMemset(b,0,bytes));
    for(int mode=0;mode<4;mode++){ int desc=mode&1, nt=mode>>1;
      for(int w=0;w<2;w++){ hipEventRecord(e0);
        for(int r=0;r<reps;r++){
          if(nt){ copyk<1><<<npieces,256>>>((d2v*)a,(d2v*)b,npieces,0); copyk<1><<<npieces,256>>>((d2v*)b,(d2v*)a,npieces,desc);} 
          else { copyk<0><<<npieces,256>>>((d2v*)a,(d2v*)b,npieces,0); copyk<0><<<npieces,256>>>((d2v*)b,(d2v*)a,npieces,desc);} }
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms,e0,e1); }
      double gbs=2.0*bytes*2*reps/1e9/(ms*1e-3);
      printf("working set %5zu MB  second pass %s  loads %s : %7.0f GB/s  (%.1f us per copy)\n",mb,desc?"descending":"ascending ",nt?"nt":"  ",gbs,ms*1e3/(2*reps));
    }
    hipFree(a); hipFree(b);
  }
  return 0; }
