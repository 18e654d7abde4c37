// Development micro-benchmark (not part of libhadi): what bandwidth do the row/column pass access
// patterns reach on their own?   hipcc -O3 --offload-arch=gfx950 tools/membench.hip -o tools/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)

// column-pass pattern: block of P waves, wave p reads rows [p*LC,(p+1)*LC) of a 64-col (VW doubles/lane) tile
template<int LC,int VW> __global__ void colpat(const double* __restrict__ in, double* __restrict__ out,int rowp,int nrows,long inst_stride,int ctiles){
  int lane=threadIdx.x&63, wave=threadIdx.x>>6;
  int inst=blockIdx.x/ctiles, ct=blockIdx.x%ctiles;
  int col=ct*64*VW+lane*VW; if(col>=rowp) col=rowp-VW;
  const double* src=in+inst*inst_stride+(long)(wave*LC)*rowp+col; double* dst=out+inst*inst_stride+(long)(wave*LC)*rowp+col;
  double y[LC][VW];
  #pragma unroll
  for(int k=0;k<LC;k++){ if(wave*LC+k<nrows){ if constexpr(VW==1) y[k][0]=src[(long)k*rowp]; else {double2 t=*(const double2*)(src+(long)k*rowp); y[k][0]=t.x;y[k][1]=t.y;} } else {y[k][0]=0; if(VW==2) y[k][VW-1]=0;} }
  double acc=0;
  #pragma unroll
  for(int k=1;k<LC;k++){ for(int v=0;v<VW;v++){ y[k][v]=fma(y[k-1][v],1e-9,y[k][v]); } }
  #pragma unroll
  for(int k=0;k<LC;k++){ if(wave*LC+k<nrows){ if constexpr(VW==1) dst[(long)k*rowp]=y[k][0]; else {double2 t; t.x=y[k][0]; t.y=y[k][1]; *(double2*)(dst+(long)k*rowp)=t;} } }
}
// row pattern: wave reads whole rows (B/2 dwordx4 per lane), R rows per wave, writes them
template<int B> __global__ void rowpat(const double* __restrict__ in,double* __restrict__ out,int rowp,int nrows,long inst_stride,int R,int ntiles){
  int lane=threadIdx.x&63, w=(blockIdx.x*blockDim.x+threadIdx.x)>>6;
  int inst=w/ntiles, tile=w%ntiles; int j0=tile*R, j1=min(j0+R,nrows);
  for(int j=j0;j<j1;j++){ const double* s=in+inst*inst_stride+(long)j*rowp; double* d=out+inst*inst_stride+(long)j*rowp;
    double2 t[B/2];
    #pragma unroll
    for(int q=0;q<B/2;q++) t[q]=*(const double2*)(s+q*128+2*lane);
    #pragma unroll
    for(int q=0;q<B/2;q++){ t[q].x*=1.0000001; *(double2*)(d+q*128+2*lane)=t[q]; } }
}
__global__ void copy4(const double2* __restrict__ in,double2* __restrict__ out,size_t n){ for(size_t i=blockIdx.x*(size_t)blockDim.x+threadIdx.x;i<n;i+=(size_t)gridDim.x*blockDim.x){double2 t=in[i]; t.x*=1.0000001; out[i]=t;} }
int main(){
  const int n_inst=256,rowp=520,nrows=257; const long stride=(long)rowp*nrows; size_t tot=(size_t)stride*n_inst;
  double *a,*b; CK(hipMalloc(&a,tot*8)); CK(hipMalloc(&b,tot*8)); CK(hipMemset(a,0,tot*8)); CK(hipMemset(b,0,tot*8));
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms; const int reps=20;
  double gb=2.0*tot*8/1e9;
  auto rep=[&](const char* name){ hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms,e0,e1); printf("%-44s %.4f ms/launch  %.0f GB/s\n",name,ms/reps,gb/(ms/reps*1e-3)); };
  for(int it=0;it<2;it++){
  hipEventRecord(e0); for(int r=0;r<reps;r++) copy4<<<4096,256>>>((double2*)a,(double2*)b,tot/2); rep("flat dwordx4 copy");
  hipEventRecord(e0); for(int r=0;r<reps;r++) colpat<65,1><<<n_inst*9,256>>>(a,b,rowp,nrows,stride,9); rep("col LC=65 P=4 dwordx2 (64 cols)");
  hipEventRecord(e0); for(int r=0;r<reps;r++) colpat<33,1><<<n_inst*9,512>>>(a,b,rowp,nrows,stride,9); rep("col LC=33 P=8 dwordx2 (64 cols)");
  hipEventRecord(e0); for(int r=0;r<reps;r++) colpat<33,2><<<n_inst*5,512>>>(a,b,rowp,nrows,stride,5); rep("col LC=33 P=8 dwordx4 (128 cols)");
  hipEventRecord(e0); for(int r=0;r<reps;r++) colpat<17,2><<<n_inst*5,1024>>>(a,b,rowp,nrows,stride,5); rep("col LC=17 P=16 dwordx4 (128 cols)");
  hipEventRecord(e0); for(int r=0;r<reps;r++) rowpat<8><<<n_inst*8*64/256,256>>>(a,b,rowp,nrows,stride,33,8); rep("row B=8 R=33 (4 waves/blk)");
  hipEventRecord(e0); for(int r=0;r<reps;r++) rowpat<8><<<n_inst*33*64/256,256>>>(a,b,rowp,nrows,stride,8,33); rep("row B=8 R=8");
  }
  return 0; }
