// probe_peak.hip -- practical FP64-MFMA ceiling of the device: every wave issues independent v_mfma_f64_16x16x4_f64
// back to back from registers (no memory traffic).  Diagnostic only (not part of the .so).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256, 2) void peak(double* out, int iters) {
  f64x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f64x4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks_per_cu, int iters) {
  double* d;
  const int nblk = 256 * blocks_per_cu;
  hipMalloc(&d, sizeof(double) * nblk * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  hipLaunchKernelGGL(peak<NACC>, dim3(nblk), dim3(256), 0, 0, d, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(peak<NACC>, dim3(nblk), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)nblk * 4 /*waves*/ * iters * NACC * 2048.0;
  printf("NACC=%2d blocks/CU=%d : %.2f ms  %.1f TFLOP/s\n", NACC, blocks_per_cu, ms, flops / ms / 1e9);
  hipFree(d);
}
int main() {
  run<16>(1, 20000);
  run<16>(2, 20000);
  run<4>(1, 80000);
  run<2>(2, 80000);
  run<16>(1, 200000);  // ~long run: sustained clock
  return 0;
}
