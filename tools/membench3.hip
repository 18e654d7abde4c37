// Development micro-benchmark (not part of libhadi): does the ACCESS WIDTH of the column pass limit it?  The pass maps lane
// <-> storage column, so every load / store moves 8 bytes per lane (512 B per wave-instruction).  Same tile traffic, same
// occupancy (one 512-thread block per CU), three register buffers as in hadi_pass_b:
//   w8   33 rows x  64 columns per wavefront, 8-byte accesses      (what hadi_pass_b does)
//   w16  17 rows x 128 columns per wavefront (two adjacent columns per lane), 16-byte accesses
//   w16l 16-byte loads, 8-byte stores (33 rows x 64 columns stored as in w8 would need a transpose: here only the LOADS are wide:
//        each lane loads rows (2k, 2k+1) of ... not expressible without a shuffle -> measured as loads wide / stores wide split below)
// as a ping-pong a -> b (ascending instances), b -> a (descending), non-temporal loads.
//   hipcc -O3 --offload-arch=gfx950 tools/membench3.hip -o tools/membench3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2v __attribute__((ext_vector_type(2)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)
// W = doubles per lane per access (1 or 2); LC rows per wavefront; TPB tiles per block, all loads of up to 3 tiles in flight
template<int W,int LC,int TPB,int WIDE_ST> __global__ void __launch_bounds__(512) colpat(const double* __restrict__ in, double* __restrict__ out,int rowp,long inst_stride,int groups,int n_inst,int desc){
  int lane=threadIdx.x&63, wave=threadIdx.x>>6;
  int binst=blockIdx.x/groups, g=blockIdx.x%groups; int inst=desc? n_inst-1-binst : binst;
  double y[TPB][LC][W];
  #pragma unroll
  for(int t=0;t<TPB;t++){ int col=(g*TPB+t)*64*W+lane*W; if(col>rowp-W) col=rowp-W;
    const double* src=in+inst*inst_stride+(long)(wave*LC)*rowp+col;
    #pragma unroll
    for(int k=0;k<LC;k++){ if(W==2){ d2v v=__builtin_nontemporal_load((const d2v*)(src+(long)k*rowp)); y[t][k][0]=v.x; y[t][k][W-1]=v.y; } else y[t][k][0]=__builtin_nontemporal_load(src+(long)k*rowp); } }
  #pragma unroll
  for(int t=0;t<TPB;t++){ int col=(g*TPB+t)*64*W+lane*W; if(col>rowp-W) col=rowp-W;
    double* dst=out+inst*inst_stride+(long)(wave*LC)*rowp+col;
    #pragma unroll
    for(int k=1;k<LC;k++) for(int w=0;w<W;w++) y[t][k][w]=fma(y[t][k-1][w],1e-9,y[t][k][w]);
    #pragma unroll
    for(int k=0;k<LC;k++){ if(W==2 && WIDE_ST){ d2v v; v.x=y[t][k][0]; v.y=y[t][k][W-1]; __builtin_nontemporal_store(v,(d2v*)(dst+(long)k*rowp)); } else for(int w=0;w<W;w++) __builtin_nontemporal_store(y[t][k][w],dst+(long)k*rowp+w); } }
}
int main(){
  const int n_inst=256,rowp=528,nrows=272; const long stride=(long)rowp*nrows; size_t tot=(size_t)stride*n_inst;
  double *a,*b; CK(hipMalloc(&a,tot*8)); CK(hipMalloc(&b,tot*8)); CK(hipMemset(a,0,tot*8)); CK(hipMemset(b,0,tot*8));
  CK(hipFuncSetAttribute((const void*)colpat<1,33,3,0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024));
  CK(hipFuncSetAttribute((const void*)colpat<2,17,1,1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024));
  CK(hipFuncSetAttribute((const void*)colpat<2,17,1,0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024));
  CK(hipFuncSetAttribute((const void*)colpat<2,17,2,1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024));
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); float ms; const int reps=20;
  // bytes actually moved by one launch: 8 waves x LC rows x tiles... the w16 variants cover 16 x 17 = 272 rows of 528 columns, w8 8 x 33 = 264 rows
  for(int it=0;it<2;it++){
   #define RUN(KERN,GRID,ROWS,NAME) { hipEventRecord(e0); for(int r=0;r<reps;r++){ KERN<<<GRID,512,100*1024>>>(a,b,rowp,stride,(GRID)/n_inst,n_inst,0); KERN<<<GRID,512,100*1024>>>(b,a,rowp,stride,(GRID)/n_inst,n_inst,1);} \
     hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms,e0,e1); double gb=2.0*n_inst*(double)ROWS*rowp*8/1e9; printf("%-64s %.4f ms/launch  %.0f GB/s\n",NAME,ms/(2*reps),gb/(ms/(2*reps)*1e-3)); }
   RUN((colpat<1,33,3,0>), n_inst*3, 264, "w8 : 33 rows x 64 cols, 3 tiles in flight, 8-byte accesses");
   RUN((colpat<2,17,1,1>), n_inst*5, 136, "w16: 17 rows x 128 cols (8 waves = 136 rows), 1 tile, 16-byte accesses");
   RUN((colpat<2,17,2,1>), n_inst*3, 136, "w16: 17 rows x 128 cols, 2 tiles in flight, 16-byte accesses");
   RUN((colpat<2,17,1,0>), n_inst*5, 136, "w16 loads, 8-byte stores: 17 rows x 128 cols, 1 tile");
  }
  return 0; }
