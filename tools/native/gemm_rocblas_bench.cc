// How fast is the vendor DGEMM on the Poisson shapes?  (diagnostic only; the product uses its own MFMA kernel)
//   hipcc --offload-arch=gfx950 -O2 tools/native/gemm_rocblas_bench.cc -lrocblas -o gpurun_out/gemm_rocblas_bench
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <cstdio>
#include <vector>
#define CK(x) do { auto e = (x); if ((int)e != 0) { printf("error %d at %s:%d\n", (int)e, __FILE__, __LINE__); return 1; } } while (0)
int main() {
  const int me = 2048, mo = 2047, my = 4095;
  const long ldx = 4112, ldy = 4112;
  double *A, *B, *C;
  CK(hipMalloc(&A, sizeof(double) * 4100 * ldx));
  CK(hipMalloc(&B, sizeof(double) * 4100 * ldx));
  CK(hipMalloc(&C, sizeof(double) * 4100 * ldy));
  std::vector<double> h((size_t)4100 * ldx);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5;
  CK(hipMemcpy(A, h.data(), sizeof(double) * 4100 * ldx, hipMemcpyHostToDevice));
  CK(hipMemcpy(B, h.data(), sizeof(double) * 4100 * ldx, hipMemcpyHostToDevice));
  rocblas_handle hd;
  CK(rocblas_create_handle(&hd));
  const double one = 1.0, zero = 0.0;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct Case { const char* name; rocblas_operation ta, tb; int m, n, k; long lda, ldb, ldc; } cases[] = {
      // row-major C[me x my] = F[me x me] * Y[my x me]^T  ==  col-major C^T = Y-as-colmajor^T * F-as-colmajor
      {"G1 (T,N) m=my n=me k=me", rocblas_operation_transpose, rocblas_operation_none, my, me, me, ldx, ldx, ldy},
      // row-major C[me x my] = F[me x me] * X[me x my]  ==  col-major C^T = X^T-as-colmajor * F-as-colmajor
      {"G2 (N,N) m=my n=me k=me", rocblas_operation_none, rocblas_operation_none, my, me, me, ldy, ldx, ldy},
      {"G2 (N,T) m=my n=me k=me", rocblas_operation_none, rocblas_operation_transpose, my, me, me, ldy, ldx, ldy},
      {"G1 odd (T,N) k=2047", rocblas_operation_transpose, rocblas_operation_none, my, mo, mo, ldx, ldx, ldy},
      {"G2 odd (N,N) k=2047", rocblas_operation_none, rocblas_operation_none, my, mo, mo, ldy, ldx, ldy},
      {"(N,N) 4096 x 2048 x 2048", rocblas_operation_none, rocblas_operation_none, 4096, 2048, 2048, ldy, ldx, ldy},
      {"square (N,N) 4096", rocblas_operation_none, rocblas_operation_none, 4096, 4096, 4096, ldx, ldx, ldy},
      {"square (T,N) 4096", rocblas_operation_transpose, rocblas_operation_none, 4096, 4096, 4096, ldx, ldx, ldy},
  };
  for (auto& c : cases) {
    for (int w = 0; w < 3; ++w)
      CK(rocblas_dgemm(hd, c.ta, c.tb, c.m, c.n, c.k, &one, B, c.lda, A, c.ldb, &zero, C, c.ldc));
    CK(hipEventRecord(e0, 0));
    const int reps = 20;
    for (int r = 0; r < reps; ++r)
      CK(rocblas_dgemm(hd, c.ta, c.tb, c.m, c.n, c.k, &one, B, c.lda, A, c.ldb, &zero, C, c.ldc));
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("%-28s %.4f ms  %.2f TFLOP/s\n", c.name, ms, 2.0 * c.m * c.n * c.k / ms / 1e9);
  }
  return 0;
}
