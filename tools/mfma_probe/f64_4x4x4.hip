// Operand / result layout of v_mfma_f64_4x4x4f64 (4 blocks of 4 x 4 x 4) found empirically: A has a single 1 at lane la,
// B holds lane + 1; the non-zero results tell which (block, i) the A lane feeds and which B lanes share its (block, k).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(int la, double* out) {
  const int l = threadIdx.x;
  const double a = l == la ? 1.0 : 0.0, b = (double)(l + 1);
  out[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
}
int main() {
  double* d;
  hipMalloc(&d, 64 * sizeof(double));
  double h[64];
  for (int la = 0; la < 64; ++la) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, la, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("A lane %2d ->", la);
    for (int l = 0; l < 64; ++l)
      if (h[l] != 0.0) printf(" D[lane %2d] = B lane %2d;", l, (int)h[l] - 1);
    printf("\n");
  }
  return 0;
}
