// calibration of the FETCH_SIZE / WRITE_SIZE counters on the access patterns of this path's gather kernels (VERDICT r03
// item 3): a 2 GB buffer (8 x the Infinity Cache) read as (a) a wide coalesced stream, 16 B per lane, (b) one 8-byte
// word per lane from a line of its own (stride 128 B), (c) 24 bytes per lane (a vertex' xyz) at scattered vertices,
// (d) 72 bytes per lane (a 3 x 3 block) at scattered blocks, and written as (e) a coalesced 16-B stream, (f) scattered
// 8-byte words, (g) scattered 24-byte vertices.  Every launch touches a KNOWN number of useful bytes; run under
// `rocprofv3 --kernel-trace --pmc FETCH_SIZE` (and WRITE_SIZE) -> counter bytes per useful byte (tools/pmc_gather.sh).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{ auto e_=(x); if((int)e_!=0){printf("fail %s -> %d\n",#x,(int)e_); exit(1);} }while(0)
__device__ __forceinline__ unsigned long long mix(unsigned long long x){ x^=x>>33; x*=0xff51afd7ed558ccdULL; x^=x>>33; x*=0xc4ceb9fe1a85ec53ULL; x^=x>>33; return x; }
__global__ __launch_bounds__(256) void cal_stream16(const double2* __restrict__ a, size_t n2, double* out){
  double acc=0; for(size_t i=(size_t)blockIdx.x*256+threadIdx.x;i<n2;i+=(size_t)gridDim.x*256){ double2 v=a[i]; acc+=v.x+v.y; }
  if(acc==1.2345) out[0]=acc; }
__global__ __launch_bounds__(256) void cal_gather8_line(const double* __restrict__ a, size_t nlines, size_t count, double* out){
  double acc=0; for(size_t i=(size_t)blockIdx.x*256+threadIdx.x;i<count;i+=(size_t)gridDim.x*256){ acc+=a[(mix(i)%nlines)*16]; }
  if(acc==1.2345) out[0]=acc; }
__global__ __launch_bounds__(256) void cal_gather24(const double* __restrict__ a, size_t nvert, size_t count, double* out){
  double acc=0; for(size_t i=(size_t)blockIdx.x*256+threadIdx.x;i<count;i+=(size_t)gridDim.x*256){ const double* p=a+3*(mix(i)%nvert); acc+=p[0]+p[1]+p[2]; }
  if(acc==1.2345) out[0]=acc; }
__global__ __launch_bounds__(256) void cal_gather72(const double* __restrict__ a, size_t nblk, size_t count, double* out){
  double acc=0; for(size_t i=(size_t)blockIdx.x*256+threadIdx.x;i<count;i+=(size_t)gridDim.x*256){ const double* p=a+9*(mix(i)%nblk);
#pragma unroll
    for(int k=0;k<9;++k) acc+=p[k]; }
  if(acc==1.2345) out[0]=acc; }
__global__ __launch_bounds__(256) void cal_wstream16(double2* __restrict__ a, size_t n2){
  for(size_t i=(size_t)blockIdx.x*256+threadIdx.x;i<n2;i+=(size_t)gridDim.x*256) a[i]=make_double2(1.0,2.0); }
__global__ __launch_bounds__(256) void cal_scatter8(double* __restrict__ a, size_t nlines, size_t count){
  for(size_t i=(size_t)blockIdx.x*256+threadIdx.x;i<count;i+=(size_t)gridDim.x*256) a[(mix(i)%nlines)*16]=1.0; }
__global__ __launch_bounds__(256) void cal_scatter24(double* __restrict__ a, size_t nvert, size_t count){
  for(size_t i=(size_t)blockIdx.x*256+threadIdx.x;i<count;i+=(size_t)gridDim.x*256){ double* p=a+3*(mix(i)%nvert); p[0]=1.0; p[1]=2.0; p[2]=3.0; } }
int main(){
  const size_t bytes=(size_t)2<<30, n=bytes/8; double* a; double* out; CK(hipMalloc(&a,bytes)); CK(hipMalloc(&out,8)); CK(hipMemset(a,0,bytes));
  const size_t count=(size_t)1<<24;   // scattered accesses per launch (16 M: 2 GB / 128 B lines, each line about once)
  const int blocks=4096;
  for(int rep=0;rep<3;++rep){
    cal_stream16<<<blocks,256>>>((const double2*)a,n/2,out);
    cal_gather8_line<<<blocks,256>>>(a,n/16,count,out);
    cal_gather24<<<blocks,256>>>(a,n/3,count,out);
    cal_gather72<<<blocks,256>>>(a,n/9,count,out);
    cal_wstream16<<<blocks,256>>>((double2*)a,n/2);
    cal_scatter8<<<blocks,256>>>(a,n/16,count);
    cal_scatter24<<<blocks,256>>>(a,n/3,count);
  }
  CK(hipDeviceSynchronize());
  printf("useful bytes per launch: stream16 %zu gather8_line %zu gather24 %zu gather72 %zu wstream16 %zu scatter8 %zu scatter24 %zu\n",
         bytes,count*8,count*24,count*72,bytes,count*8,count*24);
  return 0;
}
