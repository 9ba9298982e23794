// probe_coissue.hip -- does FP64 VALU work hide behind FP64 MFMA on gfx950?  Diagnostic only (not linked into the .so).
// Each wave runs `iters` k4-steps of the GEMM's 4 x 4 MFMA pattern (16 x v_mfma_f64_16x16x4) and, per k4-step, NV
// independent FP64 FMA chains' worth of VALU instructions (polynomial-like: what an on-the-fly exp() would issue).
// Reported: time for MFMA only, VALU only, both -- "both ~ max" means the pipes overlap, "both ~ sum" means they share.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int NV, bool MFMA, int WPB>
__global__ __launch_bounds__(256, WPB) void k(const double* __restrict__ src, double* out, int iters) {
  f64x4 acc[4][4];
  for (int a = 0; a < 4; ++a)
    for (int b = 0; b < 4; ++b) acc[a][b] = f64x4{0, 0, 0, 0};
  double fa[4], fb[4];
  const int t = threadIdx.x;
  for (int i = 0; i < 4; ++i) fa[i] = src[t + 256 * i], fb[i] = src[t + 256 * (4 + i)];
  double v[8];
  for (int i = 0; i < 8; ++i) v[i] = src[(t * 8 + i) & 2047];
  const double c0 = src[5], c1 = src[6];
  for (int it = 0; it < iters; ++it) {
    if (MFMA) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j & 7] = __builtin_fma(v[j & 7], c0, c1);   // 8 independent chains
    if (MFMA && NV > 0) {  // ask the scheduler for 1 MFMA : NV/16 VALU interleaving (one wave must overlap by itself)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, NV / 16, 0);
      }
    }
  }
  double s = 0;
  for (int a = 0; a < 4; ++a)
    for (int b = 0; b < 4; ++b) s += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, bool MFMA, int WPB>
float run(const double* src, double* d, int nblk, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NV, MFMA, WPB>), dim3(nblk), dim3(256), 0, 0, src, d, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NV, MFMA, WPB>), dim3(nblk), dim3(256), 0, 0, src, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  double *d, *src;
  hipMalloc(&d, sizeof(double) * 1024 * 256);
  hipMalloc(&src, sizeof(double) * 2048);
  double h[2048];
  unsigned long long s = 88172645463325252ULL;
  for (int i = 0; i < 2048; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0; }
  h[5] = 0.999999, h[6] = 1e-7;
  hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice);
  const int iters = 16000;
#define ROW(NV, WPB, NBLK)                                                                                          \
  {                                                                                                                 \
    const float m = run<0, true, WPB>(src, d, NBLK, iters), v = run<NV, false, WPB>(src, d, NBLK, iters),           \
                b = run<NV, true, WPB>(src, d, NBLK, iters);                                                        \
    const double tf = (double)NBLK * 4 * iters * 16 * 2048.0 / 1e9;                                                  \
    printf("blocks/CU %d  NV %3d per 16 MFMA:  mfma %.2f ms (%.1f TF)  valu %.2f ms  both %.2f ms (%.1f TF)  sum %.2f  max %.2f\n", \
           WPB, NV, m, tf / m, v, b, tf / b, m + v, m > v ? m : v);                                                 \
  }
  ROW(32, 1, 256) ROW(64, 1, 256) ROW(128, 1, 256) ROW(192, 1, 256)
  ROW(32, 2, 512) ROW(64, 2, 512) ROW(128, 2, 512) ROW(192, 2, 512)
  return 0;
}
