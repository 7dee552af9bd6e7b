// standalone harness: dense SPD solve through the dataflow kernel and through the panel path
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define TMI_CDF_STAMPS 1
#include "dense_cholesky.h"
#include "dense_cholesky_df.h"
using namespace tmi;
__global__ void fill_tiles(const double* A, const double* b, double* tiles, int n) {
  long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  long long tot = (long long)(n + 1) * n;
  if (e >= tot) return;
  int i = e / n, j = e % n;
  if (i < n && j > i) return;
  double v = i < n ? A[(size_t)i * n + j] : b[j];
  tiles[cdf::tile_index(i >> 6, j >> 6) * cdf::TILE + (size_t)(i & 63) * 64 + (j & 63)] = v;
}
extern "C" int chol_df_solve(const double* hA, const double* hb, double* hx, int n, int reps, double* ms_out, int* info) {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  cdf::Plan plan = cdf::make_plan(n, prop.multiProcessorCount);
  double *dA, *db, *dx, *tiles, *linv; int *flags, *ctrl;
  hipMalloc(&dA, (size_t)n * n * 8); hipMalloc(&db, n * 8); hipMalloc(&dx, n * 8);
  hipMalloc(&tiles, plan.tile_doubles * 8); hipMalloc(&linv, plan.linv_doubles * 8);
  hipMalloc(&flags, plan.flag_ints * 4); hipMalloc(&ctrl, 8 * 4);
  hipMemcpy(dA, hA, (size_t)n * n * 8, hipMemcpyHostToDevice); hipMemcpy(db, hb, n * 8, hipMemcpyHostToDevice);
  hipMemset(flags, 0, plan.flag_ints * 4); hipMemset(ctrl, 0, 32);
  size_t ntiles = cdf::tile_index(plan.T - 1, plan.T - 1) + 1;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    hipMemsetAsync(tiles, 0, plan.tile_doubles * 8, 0);
    long long tot = (long long)(n + 1) * n;
    hipLaunchKernelGGL(fill_tiles, dim3((tot + 255) / 256), dim3(256), 0, 0, dA, db, tiles, n);
    cdf::Args a; a.tiles = tiles; a.linv = linv; a.tflag = flags; a.dflag = flags + ntiles; a.xflag = a.dflag + plan.T;
    a.x = dx; a.ctrl = ctrl; a.singular = ctrl + 1; a.n = n; a.T = plan.T; a.epoch = r + 1; a.band = plan.band; a.team = plan.team; a.G = plan.G;
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(cdf::chol_dataflow_kernel, dim3(plan.G), dim3(256), 0, 0, a);
    hipEventRecord(e1, 0);
    hipError_t err = hipDeviceSynchronize();
    if (err != hipSuccess) { printf("hip error %s\n", hipGetErrorString(err)); return 1; }
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  hipMemcpy(hx, dx, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(info, ctrl, 8, hipMemcpyDeviceToHost);
  *ms_out = best;
  info[2] = plan.G; info[3] = plan.T;
  if (getenv("CDF_STAMPS")) {
    long long* st; hipMalloc(&st, 16 * 8); hipMemset(st, 0, 128);
    hipMemcpyToSymbol(HIP_SYMBOL(cdf::g_cdf_stamps), &st, sizeof(st));
    hipMemsetAsync(tiles, 0, plan.tile_doubles * 8, 0);
    long long tot = (long long)(n + 1) * n;
    hipLaunchKernelGGL(fill_tiles, dim3((tot + 255) / 256), dim3(256), 0, 0, dA, db, tiles, n);
    cdf::Args a; a.tiles = tiles; a.linv = linv; a.tflag = flags; a.dflag = flags + ntiles; a.xflag = a.dflag + plan.T;
    a.x = dx; a.ctrl = ctrl; a.singular = ctrl + 1; a.n = n; a.T = plan.T; a.epoch = reps + 1; a.band = plan.band; a.team = plan.team; a.G = plan.G;
    hipLaunchKernelGGL(cdf::chol_dataflow_kernel, dim3(plan.G), dim3(256), 0, 0, a);
    hipDeviceSynchronize();
    long long h[16]; hipMemcpy(h, st, 128, hipMemcpyDeviceToHost);
    printf("stamps (us since stamp 0; 100 MHz clock):");
    for (int i = 1; i < 11; ++i) printf(" [%d] %.2f", i, (h[i] - h[0]) * 0.01);
    printf("\n");
    st = nullptr; hipMemcpyToSymbol(HIP_SYMBOL(cdf::g_cdf_stamps), &st, sizeof(st));
  }
  hipFree(dA); hipFree(db); hipFree(dx); hipFree(tiles); hipFree(linv); hipFree(flags); hipFree(ctrl);
  return 0;
}
extern "C" int chol_panels_solve(const double* hA, const double* hb, double* hx, int n, int reps, double* ms_out, int* info) {
  double *dA, *dA0, *db, *dx, *tmp, *diag; int* flag;
  hipMalloc(&dA, (size_t)n * n * 8); hipMalloc(&dA0, (size_t)n * n * 8); hipMalloc(&db, n * 8); hipMalloc(&dx, n * 8); hipMalloc(&tmp, n * 8);
  hipMalloc(&diag, (size_t)kPanel * (n + kPanel) * 8); hipMalloc(&flag, 4); hipMemset(flag, 0, 4);
  hipMemcpy(dA0, hA, (size_t)n * n * 8, hipMemcpyHostToDevice); hipMemcpy(db, hb, n * 8, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    hipMemcpy(dA, dA0, (size_t)n * n * 8, hipMemcpyDeviceToDevice);
    hipEventRecord(e0, 0);
    dense_cholesky_solve(dA, n, db, dx, tmp, diag, flag, 0);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  hipMemcpy(hx, dx, n * 8, hipMemcpyDeviceToHost); hipMemcpy(info, flag, 4, hipMemcpyDeviceToHost);
  *ms_out = best;
  return 0;
}
