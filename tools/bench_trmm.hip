#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{ auto e_=(x); if((int)e_!=0){printf("fail %s -> %d\n",#x,(int)e_); exit(1);} }while(0)
int main(int argc,char**argv){
  int n = argc>1?atoi(argv[1]):1088, batch = argc>2?atoi(argv[2]):32; int lda=2176;
  rocblas_handle h; CK(rocblas_create_handle(&h));
  long stride=(long)lda*lda; double *A,*B,*C;
  CK(hipMalloc(&A,sizeof(double)*stride*batch)); CK(hipMalloc(&B,sizeof(double)*stride*batch)); CK(hipMalloc(&C,sizeof(double)*stride*batch));
  CK(hipMemset(A,0,sizeof(double)*stride*batch)); CK(hipMemset(B,0,sizeof(double)*stride*batch));
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double one=1, mone=-1, zero=0;
  auto timeit=[&](const char* name, double flops, auto fn){
    fn(); float best=1e30;
    for(int rep=0;rep<5;++rep){ CK(hipEventRecord(e0)); fn(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1)); if(ms<best)best=ms; }
    printf("%-44s n=%d batch=%d %8.3f ms  %7.2f TF (full-gemm-equivalent %7.2f)\n",name,n,batch,best,flops*batch/best/1e9, 2.0*n*n*n*batch/best/1e9);
  };
  double n3=(double)n*n*n;
  timeit("gemm TN  (m=n=k)", 2*n3, [&]{ CK(rocblas_dgemm_strided_batched(h,rocblas_operation_transpose,rocblas_operation_none,n,n,n,&one,A,lda,stride,B,lda,stride,&zero,C,n,(long)n*n,batch)); });
  timeit("trmm left upper T  out-of-place", n3, [&]{ CK(rocblas_dtrmm_strided_batched(h,rocblas_side_left,rocblas_fill_upper,rocblas_operation_transpose,rocblas_diagonal_non_unit,n,n,&one,A,lda,stride,B,lda,stride,C,n,(long)n*n,batch)); });
  timeit("trmm left upper N  out-of-place", n3, [&]{ CK(rocblas_dtrmm_strided_batched(h,rocblas_side_left,rocblas_fill_upper,rocblas_operation_none,rocblas_diagonal_non_unit,n,n,&one,A,lda,stride,B,lda,stride,C,n,(long)n*n,batch)); });
  timeit("trmm right upper N in-place", n3, [&]{ CK(rocblas_dtrmm_strided_batched(h,rocblas_side_right,rocblas_fill_upper,rocblas_operation_none,rocblas_diagonal_non_unit,n,n,&mone,A,lda,stride,B,lda,stride,B,lda,stride,batch)); });
  timeit("syrk upper T", n3, [&]{ CK(rocblas_dsyrk_strided_batched(h,rocblas_fill_upper,rocblas_operation_transpose,n,n,&mone,B,lda,stride,&one,C,lda,stride,batch)); });
  timeit("gemm TN half (m=n/2,n=n,k=n)", n3, [&]{ CK(rocblas_dgemm_strided_batched(h,rocblas_operation_transpose,rocblas_operation_none,n/2,n,n,&one,A,lda,stride,B,lda,stride,&zero,C,n,(long)n*n,batch)); });
  timeit("gemm TN quarter (m=n/2,n=n/2,k=n)", n3/2, [&]{ CK(rocblas_dgemm_strided_batched(h,rocblas_operation_transpose,rocblas_operation_none,n/2,n/2,n,&one,A,lda,stride,B,lda,stride,&zero,C,n,(long)n*n,batch)); });
  timeit("gemm NN (m=n/2,n=n,k=n/2)", n3/2, [&]{ CK(rocblas_dgemm_strided_batched(h,rocblas_operation_none,rocblas_operation_none,n/2,n,n/2,&one,A,lda,stride,B,lda,stride,&zero,C,n,(long)n*n,batch)); });
  return 0;
}
