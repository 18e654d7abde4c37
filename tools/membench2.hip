// Development micro-benchmark (not part of libhadi): the row-pass and column-pass access patterns as a ping-pong
// (a -> b ascending instances, b -> a descending), with and without non-temporal loads.  What can the memory system
// deliver for exactly the traffic of one Douglas step at 256 instances of 512x256 (2 x 277 MB)?
//   hipcc -O3 --offload-arch=gfx950 tools/membench2.hip -o tools/membench2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2v __attribute__((ext_vector_type(2)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)
template<int NT> __device__ inline double ld1(const double* p){ return NT ? __builtin_nontemporal_load(p) : *p; }
template<int NT> __device__ inline d2v ld2(const double* p){ return NT ? __builtin_nontemporal_load((const d2v*)p) : *(const d2v*)p; }
// column-pass pattern: block of P waves, wave p holds rows [p*LC,(p+1)*LC) of a 64-column tile; TPB tiles per block, double-buffered
template<int LC,int NT,int TPB> __global__ void __launch_bounds__(512) colpat(const double* __restrict__ in, double* __restrict__ out,int rowp,int nrows,long inst_stride,int groups,int n_inst,int desc){
  int lane=threadIdx.x&63, wave=threadIdx.x>>6;
  int binst=blockIdx.x/groups, g=blockIdx.x%groups; int inst=desc? n_inst-1-binst : binst;
  for(int t=0;t<TPB;t++){ int ct=g*TPB+t; int col=ct*64+lane; if(col>=rowp) col=rowp-1;
    const double* src=in+inst*inst_stride+(long)(wave*LC)*rowp+col; double* dst=out+inst*inst_stride+(long)(wave*LC)*rowp+col;
    double y[LC];
    #pragma unroll
    for(int k=0;k<LC;k++) y[k]=ld1<NT>(src+(long)k*rowp);
    #pragma unroll
    for(int k=1;k<LC;k++) y[k]=fma(y[k-1],1e-9,y[k]);
    #pragma unroll
    for(int k=0;k<LC;k++) dst[(long)k*rowp]=y[k];
  }
}
// row pattern: each wave streams R whole rows (B/2 dwordx4 per lane)
template<int B,int NT> __global__ void __launch_bounds__(256) rowpat(const double* __restrict__ in,double* __restrict__ out,int rowp,int nrows,long inst_stride,int R,int ntiles,int n_inst,int desc){
  int lane=threadIdx.x&63; long w=((long)blockIdx.x*blockDim.x+threadIdx.x)>>6;
  int binst=w/ntiles, tile=w%ntiles; int inst=desc? n_inst-1-binst : binst; if(binst>=n_inst) return; int j0=tile*R, j1=min(j0+R,nrows);
  for(int j=j0;j<j1;j++){ const double* s=in+inst*inst_stride+(long)j*rowp; double* d=out+inst*inst_stride+(long)j*rowp;
    d2v t[B/2];
    #pragma unroll
    for(int q=0;q<B/2;q++) t[q]=ld2<NT>(s+q*128+2*lane);
    #pragma unroll
    for(int q=0;q<B/2;q++){ t[q].x*=1.0000001; *(d2v*)(d+q*128+2*lane)=t[q]; } }
}
int main(){
  const int n_inst=256,rowp=520,nrows=264; const long stride=(long)rowp*nrows; size_t tot=(size_t)stride*n_inst;
  CK(hipFuncSetAttribute((const void*)rowpat<8,1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024)); CK(hipFuncSetAttribute((const void*)colpat<33,1,3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024));
  double *a,*b; CK(hipMalloc(&a,tot*8)); CK(hipMalloc(&b,tot*8)); CK(hipMemset(a,0,tot*8)); CK(hipMemset(b,0,tot*8));
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); float ms; const int reps=20;
  double gb=2.0*tot*8/1e9;  // per pass
  auto rep=[&](const char* name){ hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms,e0,e1); printf("%-58s %.4f ms/step  %.0f GB/s\n",name,ms/reps,2*gb/(ms/reps*1e-3)); };
  for(int it=0;it<2;it++){
   // one "step" = row pattern a->b ascending, then column pattern b->a (ascending or descending)
   #define STEP(NTA,NTB,DESC,NAME) hipEventRecord(e0); for(int r=0;r<reps;r++){ rowpat<8,NTA><<<n_inst*8*64/256,256>>>(a,b,rowp,nrows,stride,33,8,n_inst,0); colpat<33,NTB,3><<<n_inst*3,512>>>(b,a,rowp,nrows,stride,3,n_inst,DESC);} rep(NAME);
   STEP(0,0,0,"row asc + col asc");
   STEP(0,0,1,"row asc + col desc");
   STEP(0,1,1,"row asc + col desc, col loads nt");
   STEP(1,1,1,"row asc + col desc, all loads nt");
   STEP(1,1,0,"row asc + col asc, all loads nt");
   // each pattern alone as a ping-pong with itself
   hipEventRecord(e0); for(int r=0;r<reps;r++){ rowpat<8,1><<<n_inst*8*64/256,256>>>(a,b,rowp,nrows,stride,33,8,n_inst,0); rowpat<8,1><<<n_inst*8*64/256,256>>>(b,a,rowp,nrows,stride,33,8,n_inst,1);} rep("row + row desc, nt");
   hipEventRecord(e0); for(int r=0;r<reps;r++){ colpat<33,1,3><<<n_inst*3,512>>>(a,b,rowp,nrows,stride,3,n_inst,0); colpat<33,1,3><<<n_inst*3,512>>>(b,a,rowp,nrows,stride,3,n_inst,1);} rep("col + col desc, nt");
   hipEventRecord(e0); for(int r=0;r<reps;r++){ colpat<33,0,3><<<n_inst*3,512>>>(a,b,rowp,nrows,stride,3,n_inst,0); colpat<33,0,3><<<n_inst*3,512>>>(b,a,rowp,nrows,stride,3,n_inst,0);} rep("col + col asc");
   // the same with the occupancy of libhadi's kernels: dynamic LDS limits the row pattern to 8 waves per CU (two
   // 4-wave blocks) and the column pattern to one 8-wave block per CU
   hipEventRecord(e0); for(int r=0;r<reps;r++){ rowpat<8,1><<<n_inst*8*64/256,256,72*1024>>>(a,b,rowp,nrows,stride,33,8,n_inst,0); colpat<33,1,3><<<n_inst*3,512,100*1024>>>(b,a,rowp,nrows,stride,3,n_inst,1);} rep("row + col desc, nt, 8 waves/CU each");
   hipEventRecord(e0); for(int r=0;r<reps;r++){ rowpat<8,1><<<n_inst*8*64/256,256,72*1024>>>(a,b,rowp,nrows,stride,33,8,n_inst,0); colpat<33,1,3><<<n_inst*3,512>>>(b,a,rowp,nrows,stride,3,n_inst,1);} rep("row 8 waves/CU + col 16 waves/CU");
   hipEventRecord(e0); for(int r=0;r<reps;r++){ rowpat<8,1><<<n_inst*8*64/256,256>>>(a,b,rowp,nrows,stride,33,8,n_inst,0); colpat<33,1,3><<<n_inst*3,512,100*1024>>>(b,a,rowp,nrows,stride,3,n_inst,1);} rep("row full occupancy + col 8 waves/CU");
  }
  return 0; }
