// probe_rowpass.hip -- the PRODUCT row-pass kernels (gemm_rowpass.hip, included verbatim) under the probe harness: plain forward
// (role 1, store-only epilogue) and the full-tile Gram on random operands.  Diagnostic only.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DHM_OLD_LOOP] probe_rowpass.hip ../../hetmogp_amd/csrc/build/gemm_f64.o ../../hetmogp_amd/csrc/build/gemm_small.o
#include "../../hetmogp_amd/csrc/gemm_rowpass.hip"
#include <cstdio>
#include <vector>
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 131072, N = 1024, K = 1024;
  double *A, *B, *C;
  hipMalloc(&A, sizeof(double) * (size_t)n * K), hipMalloc(&B, sizeof(double) * (size_t)K * N), hipMalloc(&C, sizeof(double) * (size_t)n * N);
  std::vector<double> h((size_t)n * K);
  unsigned long long s = 88172645463325252ULL;
  for (auto& x : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = (double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0; }
  hipMemcpy(A, h.data(), sizeof(double) * (size_t)n * K, hipMemcpyHostToDevice);
  hipMemcpy(B, h.data(), sizeof(double) * (size_t)K * N, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  GemmArgs g;
  g.A = A, g.lda = K, g.a_kmajor = 0, g.B = B, g.ldb = N, g.b_kmajor = 1, g.C = C, g.ldc = N, g.M = n, g.N = N, g.K = K, g.role = 1;
  if (!gemm_rowpass_eligible(g)) { printf("not eligible\n"); return 1; }
  for (int rep = 0; rep < 3; ++rep) {
    launch_gemm_rowpass(g, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) launch_gemm_rowpass(g, nullptr);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 10;
    double c;
    hipMemcpy(&c, C + 12345 * (size_t)N + 100, sizeof(double), hipMemcpyDeviceToHost);
    double ref = 0;
    for (int k = 0; k < K; ++k) ref += h[(size_t)12345 * K + k] * h[(size_t)k * N + 100];
    printf("product forward role 1, n=%d: %.3f ms  %.1f TFLOP/s   check %.2e\n", n, ms, 2.0 * n * N * K / ms / 1e9, c - ref);
  }
  return 0;
}
