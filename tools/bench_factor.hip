// micro-benchmark of dense SPD factor / inverse primitives (rocSOLVER / rocBLAS), FP64
#include <hip/hip_runtime.h>
#include <rocsolver/rocsolver.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do{ auto e_=(x); if((int)e_!=0){printf("fail %s -> %d\n",#x,(int)e_); exit(1);} }while(0)
__global__ void fill_spd(double* A, int n, long stride){
  long b = blockIdx.y; int i = blockIdx.x*blockDim.x+threadIdx.x; if(i>=n*n) return;
  int r=i/n,c=i%n; double v = 1.0/(1.0+abs(r-c)); if(r==c) v+= n*0.01+2; A[b*stride+i]=v;
}
__global__ void fill_eye(double* A, int n, long stride){
  long b = blockIdx.y; int i = blockIdx.x*blockDim.x+threadIdx.x; if(i>=n*n) return;
  A[b*stride+i]= (i/n==i%n)?1.0:0.0;
}
int main(int argc,char**argv){
  int n = argc>1?atoi(argv[1]):2048, batch = argc>2?atoi(argv[2]):8;
  rocblas_handle h; CK(rocblas_create_handle(&h));
  hipStream_t st; CK(hipStreamCreate(&st)); CK(rocblas_set_stream(h,st));
  long stride=(long)n*n; double *A,*B,*C; int* info;
  CK(hipMalloc(&A,sizeof(double)*stride*batch)); CK(hipMalloc(&B,sizeof(double)*stride*batch)); CK(hipMalloc(&C,sizeof(double)*stride*batch));
  CK(hipMalloc(&info,sizeof(int)*batch));
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  dim3 g((n*n+255)/256,batch);
  auto timeit=[&](const char* name, double flops, auto fn){
    float best=1e30;
    for(int rep=0;rep<3;++rep){
      fill_spd<<<g,256,0,st>>>(A,n,stride); fill_eye<<<g,256,0,st>>>(B,n,stride);
      if(rep>=0 && name[0]=='+'){ CK(rocsolver_dpotrf_strided_batched(h,rocblas_fill_lower,n,A,n,stride,info,batch)); }
      CK(hipEventRecord(e0,st)); fn(); CK(hipEventRecord(e1,st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms,e0,e1)); if(ms<best) best=ms;
    }
    printf("%-34s n=%d batch=%d  %8.3f ms  %7.2f TFLOP/s\n",name,n,batch,best,flops*batch/best/1e9);
  };
  double n3=(double)n*n*n;
  timeit("potrf_strided_batched", n3/3, [&]{ CK(rocsolver_dpotrf_strided_batched(h,rocblas_fill_lower,n,A,n,stride,info,batch)); });
  timeit("+potri_strided_batched", 2*n3/3, [&]{ CK(rocsolver_dpotri_strided_batched(h,rocblas_fill_lower,n,A,n,stride,info,batch)); });
  timeit("+potrs(identity)", 2*n3, [&]{ CK(rocsolver_dpotrs_strided_batched(h,rocblas_fill_lower,n,n,A,n,stride,B,n,stride,batch)); });
  timeit("+trtri_strided_batched", n3/3, [&]{ CK(rocsolver_dtrtri_strided_batched(h,rocblas_fill_lower,rocblas_diagonal_non_unit,n,A,n,stride,info,batch)); });
  double one=1, zero=0;
  timeit("+trsm left lower (identity rhs)", n3, [&]{ CK(rocblas_dtrsm_strided_batched(h,rocblas_side_left,rocblas_fill_lower,rocblas_operation_none,rocblas_diagonal_non_unit,n,n,&one,A,n,stride,B,n,stride,batch)); });
  timeit("gemm_strided_batched NT", 2*n3, [&]{ CK(rocblas_dgemm_strided_batched(h,rocblas_operation_none,rocblas_operation_transpose,n,n,n,&one,A,n,stride,A,n,stride,&zero,C,n,stride,batch)); });
  timeit("syrk_strided_batched", n3, [&]{ CK(rocblas_dsyrk_strided_batched(h,rocblas_fill_lower,rocblas_operation_transpose,n,n,&one,A,n,stride,&zero,C,n,stride,batch)); });
  // per-matrix potrf on one stream (non batched)
  timeit("potrf loop (non-batched)", n3/3, [&]{ for(int b=0;b<batch;++b) CK(rocsolver_dpotrf(h,rocblas_fill_lower,n,A+b*stride,n,info+b)); });
  // smaller sizes
  return 0;
}
