// ceiling probe: pure streaming FP64 read (sum) over a 1.2 GB buffer, several launch shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{ auto e_=(x); if((int)e_!=0){printf("fail %s -> %d\n",#x,(int)e_); exit(1);} }while(0)
template<int UNROLL>
__global__ __launch_bounds__(256) void stream_sum(const double2* __restrict__ a, size_t n2, double* out){
  size_t i = (size_t)blockIdx.x*256*UNROLL + threadIdx.x; double acc=0;
  size_t stride=(size_t)gridDim.x*256*UNROLL;
  for(; i + 256*(UNROLL-1) < n2; i+=stride){
    double2 v[UNROLL];
#pragma unroll
    for(int u=0;u<UNROLL;++u) v[u]=a[i+256*u];
#pragma unroll
    for(int u=0;u<UNROLL;++u) acc+=v[u].x+v[u].y;
  }
  if(acc==1.2345) out[0]=acc;
}
int main(){
  size_t bytes = (size_t)32*2176*2176*8; size_t n2=bytes/16;
  double2* a; double* out; CK(hipMalloc(&a,bytes)); CK(hipMalloc(&out,8)); CK(hipMemset(a,0,bytes));
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run=[&](const char* name, auto kern, int blocks){
    kern<<<blocks,256>>>(a,n2,out);
    CK(hipEventRecord(e0)); for(int r=0;r<20;++r) kern<<<blocks,256>>>(a,n2,out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms,e0,e1)); ms/=20;
    printf("%-28s blocks=%6d  %.4f ms  %.1f GB/s\n",name,blocks,ms,bytes/ms/1e6);
  };
  for(int blocks: {512,1024,2048,4096,8192,32768}){
    run("unroll4", stream_sum<4>, blocks);
    run("unroll8", stream_sum<8>, blocks);
    run("unroll16", stream_sum<16>, blocks);
  }
  return 0;
}
