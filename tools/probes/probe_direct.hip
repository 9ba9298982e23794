// probe_direct.hip -- experiment: LDS-free FP64-MFMA GEMM (fragments loaded global -> registers, no barriers).
// C[n x N] = A[n x K] (row-major) * B[K x N] (row-major).  Diagnostic only (not part of the .so).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

template <int WPS>
__global__ __launch_bounds__(256, WPS) void gemm_direct(const double* __restrict__ A, const double* __restrict__ B,
                                                      double* __restrict__ C, int M, int N, int K, int tiles_n, int ntiles) {
  int v = blockIdx.x;
  if ((ntiles & 7) == 0) {
    const int cpx = ntiles >> 3;
    v = (v & 7) * cpx + (v >> 3);
  }
  const int ti = v / tiles_n, tj = v - ti * tiles_n;
  const int i0 = ti * 128, j0 = tj * 128;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1, lr = lane & 15, lk = lane >> 4;
  f64x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f64x4{0, 0, 0, 0};
  // lane (lr, lk) of MFMA kk consumes k = k0 + 2*lk + kk  (A: 16 contiguous bytes per row tile; B: one row of B per kk)
  const double* pa = A + (long long)(i0 + wm * 64 + lr) * K + 2 * lk;
  const double* pb = B + (long long)(2 * lk) * N + j0 + wn * 64 + lr;
  f64x2 fa0[4], fa1[4];
  double fb0[2][4], fb1[2][4];
  auto load = [&](int k0, f64x2 (&fa)[4], double (&fb)[2][4]) {
#pragma unroll
    for (int a = 0; a < 4; ++a) fa[a] = *reinterpret_cast<const f64x2*>(pa + (long long)a * 16 * K + k0);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int b = 0; b < 4; ++b) fb[kk][b] = pb[(long long)(k0 + kk) * N + b * 16];
  };
  auto compute = [&](const f64x2 (&fa)[4], const double (&fb)[2][4]) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(kk ? fa[a].y : fa[a].x, fb[kk][b], acc[a][b], 0, 0, 0);
  };
  load(0, fa0, fb0);
  for (int k0 = 0; k0 < K; k0 += 16) {
    if (k0 + 8 < K) load(k0 + 8, fa1, fb1);
    compute(fa0, fb0);
    if (k0 + 16 < K) load(k0 + 16, fa0, fb0);
    if (k0 + 8 < K) compute(fa1, fb1);
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double* crow = C + (long long)(i0 + wm * 64 + a * 16 + 4 * r + lk) * N + j0 + wn * 64 + lr;
#pragma unroll
      for (int b = 0; b < 4; ++b) crow[b * 16] = acc[a][b][r];
    }
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 131072, N = 1024, K = 1024;
  std::vector<double> hA((size_t)n * K), hB((size_t)K * N);
  unsigned long long s = 88172645463325252ULL;
  auto rnd = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0; };
  for (auto& x : hA) x = rnd();
  for (auto& x : hB) x = rnd();
  double *A, *B, *C;
  hipMalloc(&A, sizeof(double) * hA.size()), hipMalloc(&B, sizeof(double) * hB.size()), hipMalloc(&C, sizeof(double) * (size_t)n * N);
  hipMemcpy(A, hA.data(), sizeof(double) * hA.size(), hipMemcpyHostToDevice);
  hipMemcpy(B, hB.data(), sizeof(double) * hB.size(), hipMemcpyHostToDevice);
  const int tiles_n = N / 128, ntiles = (n / 128) * tiles_n;
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  for (int wps = 1; wps <= 2; ++wps) {
    auto run = [&] {
      if (wps == 1) hipLaunchKernelGGL(gemm_direct<1>, dim3(ntiles), dim3(256), 0, 0, A, B, C, n, N, K, tiles_n, ntiles);
      else hipLaunchKernelGGL(gemm_direct<2>, dim3(ntiles), dim3(256), 0, 0, A, B, C, n, N, K, tiles_n, ntiles);
    };
    run();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) run();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    std::vector<double> hC(4 * N);
    hipMemcpy(hC.data(), C + (size_t)777 * N, sizeof(double) * N, hipMemcpyDeviceToHost);
    double err = 0;
    for (int j = 0; j < N; j += 97) {
      double r = 0;
      for (int k = 0; k < K; ++k) r += hA[(size_t)777 * K + k] * hB[(size_t)k * N + j];
      err = fmax(err, fabs(r - hC[j]));
    }
    printf("launch_bounds(256,%d): n=%d  %.3f ms  %.1f TFLOP/s  (%.1f%% of 78.6)  max|err|=%.2e\n", wps, n, ms,
           2.0 * n * N * K / ms / 1e9, 100.0 * 2.0 * n * N * K / ms / 1e9 / 78.6, err);
  }
  return 0;
}
