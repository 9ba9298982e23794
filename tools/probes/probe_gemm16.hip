// probe_gemm16.hip -- FP64-MFMA GEMM main loop with SIXTEEN waves per 128 x 128 block (wave tile 32 x 32, <= 64 VGPRs, two
// blocks per CU = EIGHT waves per SIMD) against the product layout (eight waves, wave tile 64 x 32, four waves per SIMD),
// both with the stage-first software pipeline of gemm_rowpass.hip.  Diagnostic only (not linked).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
constexpr int BM = 128, BN = 128, BK = 16, KM_LD = 144, RM_LD = 18, TILE = BK * KM_LD;

// W = 8, WGN = 4: wave grid 2 x 4, wave tile 64 x 32 (the product layout);  W = 16, WGN = 4: 4 x 4, wave tile 32 x 32;
// W = 8, WGN = 1: wave grid 8 x 1, wave tile 16 x 128 (a wave owns complete rows of the tile: one row-statistics partial
// per column tile instead of four -- at the price of 9 instead of 6 fragment reads per 8 MFMAs)
template <int W, int WGN>
__global__ __launch_bounds__(W * 64, W / 2) void gemm(const double* __restrict__ A, const double* __restrict__ B,
                                                      double* __restrict__ C, int n, int N, int K) {
  constexpr int NT = W * 64, SB = 8 / WGN, SA = 8 / (W / WGN), PER = 2048 / NT;
  static_assert(SA * 16 * (W / WGN) == 128 && SB * 16 * WGN == 128, "wave tiles cover the block tile");
  __shared__ __attribute__((aligned(16))) double la[2][TILE];
  __shared__ __attribute__((aligned(16))) double lb[2][TILE];
  const int tiles_n = N / BN;
  int v = blockIdx.x;
  const int ntiles = (n / BM) * tiles_n;
  if ((ntiles & 7) == 0) { const int cpx = ntiles >> 3; v = (v & 7) * cpx + (v >> 3); }
  const int ti = v / tiles_n, tj = v - ti * tiles_n, i0 = ti * BM, j0 = tj * BN;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, lr = lane & 15, lk = lane >> 4;
  const int wm = w / WGN, wn = w % WGN;
  f64x4 acc[SA][SB];
#pragma unroll
  for (int a = 0; a < SA; ++a)
#pragma unroll
    for (int b = 0; b < SB; ++b) acc[a][b] = f64x4{0, 0, 0, 0};
  double ra[PER], rb[PER];
  const int ar = t / (16 / PER), ak = (t % (16 / PER)) * PER;     // A row-major: row, PER consecutive k
  const int bk = t / (NT / 16), bc = (t % (NT / 16)) * PER;       // B k-major: k row, PER consecutive columns
  auto load = [&](int k0) {
    const double* pa = A + (long long)(i0 + ar) * K + k0 + ak;
    const double* pb = B + (long long)(k0 + bk) * N + j0 + bc;
#pragma unroll
    for (int i = 0; i < PER / 2; ++i) {
      f64x2 x = *reinterpret_cast<const f64x2*>(pa + 2 * i), y = *reinterpret_cast<const f64x2*>(pb + 2 * i);
      ra[2 * i] = x.x, ra[2 * i + 1] = x.y, rb[2 * i] = y.x, rb[2 * i + 1] = y.y;
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PER / 2; ++i) {
      *reinterpret_cast<f64x2*>(&la[buf][ar * RM_LD + ak + 2 * i]) = f64x2{ra[2 * i], ra[2 * i + 1]};
      *reinterpret_cast<f64x2*>(&lb[buf][bk * KM_LD + bc + 2 * i]) = f64x2{rb[2 * i], rb[2 * i + 1]};
    }
  };
  int cur = 0;
  load(0);
  stage(0);
  if (BK < K) load(BK);
  __syncthreads();
  for (int k0 = 0; k0 < K; k0 += BK) {
    if (k0 + BK < K) stage(cur ^ 1);
    if (k0 + 2 * BK < K) load(k0 + 2 * BK);
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      double fa[SA], fb[SB];
#pragma unroll
      for (int i = 0; i < SA; ++i) fa[i] = la[cur][(wm * 16 * SA + i * 16 + lr) * RM_LD + kk * 4 + lk];
#pragma unroll
      for (int i = 0; i < SB; ++i) fb[i] = lb[cur][(kk * 4 + lk) * KM_LD + wn * (16 * SB) + i * 16 + lr];
#pragma unroll
      for (int a = 0; a < SA; ++a)
#pragma unroll
        for (int b = 0; b < SB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
    __syncthreads();
    cur ^= 1;
  }
#pragma unroll
  for (int a = 0; a < SA; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double* crow = C + (long long)(i0 + wm * 16 * SA + a * 16 + 4 * r + lk) * N + j0 + wn * (16 * SB);
#pragma unroll
      for (int b = 0; b < SB; ++b) crow[b * 16 + lr] = acc[a][b][r];
    }
}

int main() {
  const int n = 131072, N = 1024, K = 1024;
  double *A, *B, *C;
  hipMalloc(&A, sizeof(double) * (size_t)n * K), hipMalloc(&B, sizeof(double) * (size_t)K * N), hipMalloc(&C, sizeof(double) * (size_t)n * N);
  std::vector<double> h((size_t)n * K);
  unsigned long long s = 88172645463325252ULL;
  for (auto& x : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = (double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0; }
  hipMemcpy(A, h.data(), sizeof(double) * (size_t)n * K, hipMemcpyHostToDevice);
  hipMemcpy(B, h.data(), sizeof(double) * (size_t)K * N, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const int blocks = (n / BM) * (N / BN);
  for (int rep = 0; rep < 2; ++rep)
    for (int variant = 0; variant < 3; ++variant) {
      auto run = [&] {
        if (variant == 0) hipLaunchKernelGGL((gemm<8, 4>), dim3(blocks), dim3(512), 0, 0, A, B, C, n, N, K);
        else if (variant == 1) hipLaunchKernelGGL((gemm<16, 4>), dim3(blocks), dim3(1024), 0, 0, A, B, C, n, N, K);
        else hipLaunchKernelGGL((gemm<8, 1>), dim3(blocks), dim3(512), 0, 0, A, B, C, n, N, K);
      };
      run();
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int i = 0; i < 10; ++i) run();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      ms /= 10;
      double c0;
      hipMemcpy(&c0, C + 12345 * (size_t)N + 100, sizeof(double), hipMemcpyDeviceToHost);
      double ref = 0;  // spot check of C[12345][100]
      for (int k = 0; k < K; ++k) ref += h[(size_t)12345 * K + k] * h[(size_t)k * N + 100];
      printf("%2d waves/block, %d waves/SIMD, wave tile %s, stage-first: %.3f ms  %.1f TFLOP/s   check %.3e\n", variant == 1 ? 16 : 8,
             variant == 1 ? 8 : 4, variant == 0 ? "64x32" : (variant == 1 ? "32x32" : "16x128"), ms, 2.0 * n * N * K / ms / 1e9, c0 - ref);
    }
  return 0;
}
