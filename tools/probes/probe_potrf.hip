// Probe: phase timing of potrf_step_kernel (compile with -DPOTRF_STAMPS) and wall time of launch_potrf_batched.
#include "common.h"
#include <cstdio>
#include <vector>
#include <cmath>
#include <cstring>
extern long long* g_potrf_stamps;
int main(int argc, char** argv) {
  const int Q = 3, M = argc > 1 ? atoi(argv[1]) : 1024;
  const long long MM = (long long)M * M;
  std::vector<double> h(Q * MM);
  for (int q = 0; q < Q; ++q)
    for (int i = 0; i < M; ++i)
      for (int j = 0; j < M; ++j) h[q * MM + i * M + j] = std::exp(-0.5 * (i - j) * (i - j) / 4.0) + (i == j ? 1e-3 : 0.0);
  double *A, *A0, *scr; int* info; long long* stamps;
  HIP_TRY(hipMalloc(&A, sizeof(double) * Q * MM)); HIP_TRY(hipMalloc(&A0, sizeof(double) * Q * MM));
  HIP_TRY(hipMalloc(&scr, sizeof(double) * Q * MM)); HIP_TRY(hipMalloc(&info, sizeof(int) * Q));
  HIP_TRY(hipMalloc(&stamps, sizeof(long long) * 16)); HIP_TRY(hipMemset(stamps, 0, sizeof(long long) * 16));
  HIP_TRY(hipMemcpy(A0, h.data(), sizeof(double) * Q * MM, hipMemcpyHostToDevice));
  g_potrf_stamps = stamps;
  hipEvent_t e0, e1; HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
  for (int it = 0; it < 5; ++it) {
    HIP_TRY(hipMemcpy(A, A0, sizeof(double) * Q * MM, hipMemcpyDeviceToDevice));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipEventRecord(e0, 0));
    launch_potrf_batched(A, Q, M, info, scr, 0);
    HIP_TRY(hipEventRecord(e1, 0));
    HIP_TRY(hipEventSynchronize(e1));
    float ms; HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    long long st[16]; HIP_TRY(hipMemcpy(st, stamps, sizeof st, hipMemcpyDeviceToHost));
    printf("potrf Q=%d M=%d: %.3f ms;  stamps 1..7 (clock64 ticks since kernel start; block 0 of the MIDDLE panel launch; 6 = look-ahead block done):", Q, M, ms);
    for (int i = 1; i < 8; ++i) printf(" %lld", st[i] - st[0]);
    printf("\n");
  }
  // bit-level fingerprint of the factor (lower triangles): A/B runs of kernel variants must print the same value
  HIP_TRY(hipMemcpy(h.data(), A, sizeof(double) * Q * MM, hipMemcpyDeviceToHost));
  unsigned long long fp = 1469598103934665603ull;
  for (int q = 0; q < Q; ++q)
    for (int i = 0; i < M; ++i)
      for (int j = 0; j <= i; ++j) {
        unsigned long long b;
        memcpy(&b, &h[q * MM + i * M + j], 8);
        fp = (fp ^ b) * 1099511628211ull;
      }
  printf("factor fingerprint %016llx\n", fp);
  int hi[3]; HIP_TRY(hipMemcpy(hi, info, sizeof hi, hipMemcpyDeviceToHost));
  printf("info %d %d %d\n", hi[0], hi[1], hi[2]);
  return 0;
}
