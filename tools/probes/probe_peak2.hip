// probe_peak2.hip -- FP64-MFMA issue ceiling with the REAL operand pattern of the GEMM wave tile: 4 A fragments x 4 B
// fragments -> 16 accumulators, fragments refreshed from memory-resident values every "k-step".  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256, 2) void peak(const double* __restrict__ src, double* out, int iters) {
  f64x4 acc[4][4];
  for (int a = 0; a < 4; ++a)
    for (int b = 0; b < 4; ++b) acc[a][b] = f64x4{0, 0, 0, 0};
  double fa[4], fb[4];
  const int t = threadIdx.x;
  for (int i = 0; i < 4; ++i) fa[i] = src[t + 256 * i], fb[i] = src[t + 256 * (4 + i)];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (MODE == 1) {  // perturb the fragments with cheap VALU so that operand values change like real data
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = fa[i] * 1.0000001 + 1e-9, fb[i] = fb[i] * 0.9999999 - 1e-9;
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
  }
  double s = 0;
  for (int a = 0; a < 4; ++a)
    for (int b = 0; b < 4; ++b) s += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  double *d, *src;
  const int nblk = 512;
  hipMalloc(&d, sizeof(double) * nblk * 256);
  hipMalloc(&src, sizeof(double) * 2048);
  double h[2048];
  unsigned long long s = 88172645463325252ULL;
  for (int i = 0; i < 2048; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0; }
  hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode) {
    const int iters = 4000;
    auto run = [&](int it) {
      if (mode == 0) hipLaunchKernelGGL(peak<0>, dim3(nblk), dim3(256), 0, 0, src, d, it);
      else hipLaunchKernelGGL(peak<1>, dim3(nblk), dim3(256), 0, 0, src, d, it);
    };
    run(10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    run(iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)nblk * 4 * iters * 64 * 2048.0;
    printf("mode %d (4x4 fragment pattern%s): %.2f ms  %.1f TFLOP/s\n", mode, mode ? ", operands perturbed by VALU" : ", random constants", ms, flops / ms / 1e9);
  }
  return 0;
}
