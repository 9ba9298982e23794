// probe_gemm8.hip -- does the FP64-MFMA GEMM main loop gain from FOUR waves per SIMD?  Diagnostic only (not linked).
// Same block tile (128 x 128 x 16), LDS images and double buffering as hetmogp_amd/csrc/gemm_f64.hip (forward shape:
// A row-major n x K, B k-major K x N), full tiles only, two variants:
//   W = 4 : 256 threads, wave tile 64 x 64 (16 accumulators, 2 blocks / CU = 2 waves / SIMD)   -- the product kernel
//   W = 8 : 512 threads, wave tile 64 x 32 ( 8 accumulators, 2 blocks / CU = 4 waves / SIMD)
// probe_coissue.hip shows ONE wave per SIMD reaches only half the FP64-MFMA rate and two reach all of it, i.e. with two
// waves per SIMD any stall (barrier, waitcnt) of one of them idles half the pipe.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
constexpr int BM = 128, BN = 128, BK = 16, KM_LD = 144, RM_LD = 18, TILE = BK * KM_LD;

template <int W, bool SWAP = false, bool FRAGPF = false>
__global__ __launch_bounds__(W * 64, W / 2) void gemm(const double* __restrict__ A, const double* __restrict__ B,
                                                  double* __restrict__ C, int n, int N, int K) {
  constexpr int NT = W * 64, NBSUB = (W == 4) ? 4 : 2, PER = 2048 / NT;  // doubles per thread per operand tile
  __shared__ __attribute__((aligned(16))) double la[2][TILE];
  __shared__ __attribute__((aligned(16))) double lb[2][TILE];
  const int tiles_n = N / BN;
  int v = blockIdx.x;
  const int ntiles = (n / BM) * tiles_n;
  if ((ntiles & 7) == 0) { const int cpx = ntiles >> 3; v = (v & 7) * cpx + (v >> 3); }
  const int ti = v / tiles_n, tj = v - ti * tiles_n, i0 = ti * BM, j0 = tj * BN;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, lr = lane & 15, lk = lane >> 4;
  const int wm = (W == 4) ? (w >> 1) : (w >> 2), wn = (W == 4) ? (w & 1) : (w & 3);
  f64x4 acc[4][NBSUB];
  for (int a = 0; a < 4; ++a)
    for (int b = 0; b < NBSUB; ++b) acc[a][b] = f64x4{0, 0, 0, 0};
  double ra[PER], rb[PER];
  // A row-major: thread -> row r, PER consecutive k;  B k-major: thread -> k row, column pairs
  const int ar = t / (16 / PER), ak = (t % (16 / PER)) * PER;
  const int bk = t / (NT / 16), bc = (t % (NT / 16)) * 2;  // W=4: 16 lanes x 2 cols, 4 groups of 32; W=8: 32 lanes x 2 cols, 2 groups of 64
  auto load = [&](int k0) {
    const double* pa = A + (long long)(i0 + ar) * K + k0 + ak;
#pragma unroll
    for (int i = 0; i < PER / 2; ++i) { f64x2 x = *reinterpret_cast<const f64x2*>(pa + 2 * i); ra[2 * i] = x.x, ra[2 * i + 1] = x.y; }
    const double* pb = B + (long long)(k0 + bk) * N + j0 + bc;
#pragma unroll
    for (int j = 0; j < PER / 2; ++j) { f64x2 x = *reinterpret_cast<const f64x2*>(pb + j * (NT / 8)); rb[2 * j] = x.x, rb[2 * j + 1] = x.y; }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PER / 2; ++i) *reinterpret_cast<f64x2*>(&la[buf][ar * RM_LD + ak + 2 * i]) = f64x2{ra[2 * i], ra[2 * i + 1]};
#pragma unroll
    for (int j = 0; j < PER / 2; ++j) *reinterpret_cast<f64x2*>(&lb[buf][bk * KM_LD + bc + j * (NT / 8)]) = f64x2{rb[2 * j], rb[2 * j + 1]};
  };
  int cur = 0;
  load(0);
  stage(0);
  __syncthreads();
  for (int k0 = 0; k0 < K; k0 += BK) {
    const bool more = k0 + BK < K;
    if (more) load(k0 + BK);
    if (!FRAGPF) {
#pragma unroll
      for (int kk = 0; kk < BK / 4; ++kk) {
        double fa[4], fb[NBSUB];
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = la[cur][(wm * 64 + i * 16 + lr) * RM_LD + kk * 4 + lk];
#pragma unroll
        for (int i = 0; i < NBSUB; ++i) fb[i] = lb[cur][(kk * 4 + lk) * KM_LD + wn * (16 * NBSUB) + i * 16 + (SWAP ? 4 * (lr & 3) + (lr >> 2) : lr)];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < NBSUB; ++b)
            acc[a][b] = SWAP ? __builtin_amdgcn_mfma_f64_16x16x4f64(fb[b], fa[a], acc[a][b], 0, 0, 0)
                             : __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], fb[b], acc[a][b], 0, 0, 0);
      }
    } else {  // fragments of k4-step kk+1 are read from LDS before the MFMAs of kk are issued (two register sets)
      double fa[2][4], fb[2][NBSUB];
      auto frag = [&](int kk, int s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[s][i] = la[cur][(wm * 64 + i * 16 + lr) * RM_LD + kk * 4 + lk];
#pragma unroll
        for (int i = 0; i < NBSUB; ++i) fb[s][i] = lb[cur][(kk * 4 + lk) * KM_LD + wn * (16 * NBSUB) + i * 16 + (SWAP ? 4 * (lr & 3) + (lr >> 2) : lr)];
      };
      frag(0, 0);
#pragma unroll
      for (int kk = 0; kk < BK / 4; ++kk) {
        if (kk + 1 < BK / 4) frag(kk + 1, (kk + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);   // keep the prefetch in front of this k4-step's MFMAs
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < NBSUB; ++b)
            acc[a][b] = SWAP ? __builtin_amdgcn_mfma_f64_16x16x4f64(fb[kk & 1][b], fa[kk & 1][a], acc[a][b], 0, 0, 0)
                             : __builtin_amdgcn_mfma_f64_16x16x4f64(fa[kk & 1][a], fb[kk & 1][b], acc[a][b], 0, 0, 0);
      }
    }
    if (more) stage(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
  if (SWAP) {  // transposed sub-tiles: lane (lr, lk) holds row wm*64 + a*16 + lr, 4 adjacent columns per sub-tile
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      double* crow = C + (long long)(i0 + wm * 64 + a * 16 + lr) * N + j0 + wn * (16 * NBSUB) + 4 * lk;
#pragma unroll
      for (int b = 0; b < NBSUB; ++b) {
        *reinterpret_cast<f64x2*>(crow + b * 16) = f64x2{acc[a][b][0], acc[a][b][1]};
        *reinterpret_cast<f64x2*>(crow + b * 16 + 2) = f64x2{acc[a][b][2], acc[a][b][3]};
      }
    }
    return;
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double* crow = C + (long long)(i0 + wm * 64 + a * 16 + 4 * r + lk) * N + j0 + wn * (16 * NBSUB);
#pragma unroll
      for (int b = 0; b < NBSUB; ++b) crow[b * 16 + lr] = acc[a][b][r];
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
  for (int variant = 0; variant < 5; ++variant) {
    auto run = [&] {
      if (variant == 0) hipLaunchKernelGGL(gemm<4>, dim3(blocks), dim3(256), 0, 0, A, B, C, n, N, K);
      else if (variant == 1) hipLaunchKernelGGL(gemm<8>, dim3(blocks), dim3(512), 0, 0, A, B, C, n, N, K);
      else if (variant == 2) hipLaunchKernelGGL((gemm<8, true>), dim3(blocks), dim3(512), 0, 0, A, B, C, n, N, K);
      else if (variant == 3) hipLaunchKernelGGL((gemm<8, false, true>), dim3(blocks), dim3(512), 0, 0, A, B, C, n, N, K);
      else hipLaunchKernelGGL((gemm<8, true, true>), dim3(blocks), dim3(512), 0, 0, A, B, C, n, N, K);
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
    std::vector<double> c(16);
    hipMemcpy(c.data(), C + 12345 * (size_t)N + 100, sizeof(double) * 16, hipMemcpyDeviceToHost);
    double ref = 0;  // spot check of C[12345][100]
    for (int k = 0; k < K; ++k) ref += h[(size_t)12345 * K + k] * h[(size_t)k * N + 100];
    printf("W=%d waves/block (%d waves/SIMD)%s: %.3f ms  %.1f TFLOP/s   check %.3e\n", variant ? 8 : 4, variant ? 4 : 2,
           variant == 2 ? " transposed accumulators" : (variant == 3 ? " fragment prefetch" : (variant == 4 ? " transposed + fragment prefetch" : "")), ms, 2.0 * n * N * K / ms / 1e9, c[0] - ref);
  }
  return 0;
}
