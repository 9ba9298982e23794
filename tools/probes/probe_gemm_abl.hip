// probe_gemm_abl.hip -- ABLATION of the 8-wave stage-first FP64-MFMA main loop (the forward's loop of gemm_rowpass.hip): which
// part of a k-step costs the ~10 % between the bare loop (~70 TFLOP/s) and the pure-MFMA rate (~77.5)?  Diagnostic only.
// Flags (bit mask): 1 = no barrier in the loop (racy: timing only), 2 = no global loads (registers reused), 4 = no ds_write
// (LDS staged once), 8 = no ds_read (fragments stay in registers), 16 = fragment reads as one ds_read_b64 each via opaque
// offsets, 32 = s_setprio 3 around the MFMA block.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
constexpr int BM = 128, BN = 128, BK = 16, KM_LD = 144, RM_LD = 18, TILE = BK * KM_LD, W = 8, NT = 512;

template <int F>
__global__ __launch_bounds__(NT, 4) void gemm(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C,
                                              int n, int N, int K) {
  __shared__ __attribute__((aligned(16))) double la[2][TILE];
  __shared__ __attribute__((aligned(16))) double lb[2][TILE];
  __shared__ unsigned ctr[2];     // flag 64: [0] = tiles staged (x8 waves), [1] = tiles consumed (x8 waves); monotonic
  if (F & 64) {
    if (threadIdx.x < 2) ctr[threadIdx.x] = 0;
  }
  const int tiles_n = N / BN;
  int v = blockIdx.x;
  const int ntiles = (n / BM) * tiles_n;
  if ((ntiles & 7) == 0) { const int cpx = ntiles >> 3; v = (v & 7) * cpx + (v >> 3); }
  const int ti = v / tiles_n, tj = v - ti * tiles_n, i0 = ti * BM, j0 = tj * BN;
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), lr = lane & 15, lk = lane >> 4;
  const int wm = w >> 2, wn = w & 3;
  f64x4 acc[4][2];
  for (int a = 0; a < 4; ++a)
    for (int b = 0; b < 2; ++b) acc[a][b] = f64x4{0, 0, 0, 0};
  double ra[4], rb[4];
  const int ar = t >> 2, ak = (t & 3) * 4, bk = t >> 5, bc = (t & 31) * 2;
  const double* pa = A + (long long)(i0 + ar) * K + ak;
  const double* pb = B + (long long)bk * N + j0 + bc;
  auto load = [&](int k0) {
    const f64x2 x0 = *reinterpret_cast<const f64x2*>(pa + k0), x1 = *reinterpret_cast<const f64x2*>(pa + k0 + 2);
    ra[0] = x0.x, ra[1] = x0.y, ra[2] = x1.x, ra[3] = x1.y;
    const double* q = pb + (long long)k0 * N;
    const f64x2 y0 = *reinterpret_cast<const f64x2*>(q), y1 = *reinterpret_cast<const f64x2*>(q + 64);
    rb[0] = y0.x, rb[1] = y0.y, rb[2] = y1.x, rb[3] = y1.y;
  };
  auto stage = [&](int buf) {
    double* sa = &la[buf][ar * RM_LD + ak];
    *reinterpret_cast<f64x2*>(sa) = f64x2{ra[0], ra[1]};
    *reinterpret_cast<f64x2*>(sa + 2) = f64x2{ra[2], ra[3]};
    double* sb = &lb[buf][bk * KM_LD + bc];
    *reinterpret_cast<f64x2*>(sb) = f64x2{rb[0], rb[1]};
    *reinterpret_cast<f64x2*>(sb + 64) = f64x2{rb[2], rb[3]};
  };
  double fa[4] = {1.0, 2.0, 3.0, 4.0}, fb[2] = {0.5, 0.25};
  auto mma = [&](int buf) {
    if (F & 32) __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      if (!(F & 8)) {
        const double* fpa = &la[buf][(wm * 64 + lr) * RM_LD + kk * 4 + lk];
        const double* fpb = &lb[buf][(kk * 4 + lk) * KM_LD + wn * 32 + lr];
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = fpa[i * 16 * RM_LD];
#pragma unroll
        for (int i = 0; i < 2; ++i) fb[i] = fpb[i * 16];
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
    if (F & 32) __builtin_amdgcn_s_setprio(0);
  };
  auto arrive = [&](int which) {      // this wave's LDS operations so far have completed (in-order LDS pipeline)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(&ctr[which], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  auto wait_ge = [&](int which, unsigned target) {
    while (__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&ctr[which], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < (int)target)
      __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
  };
  int cur = 0;
  load(0);
  stage(0);
  if (F & 4) stage(1);
  load(BK);
  __syncthreads();
  if (F & 128) {
    // fragments one k4-step ahead, carried across the barrier: the first fragments of tile k+1 are read right behind the barrier,
    // in front of the last 8 MFMAs of tile k -- no wave ever waits for LDS without 8 MFMAs of its own in flight
    double ga[2][4], gb[2][2];
    auto frag = [&](int buf, int kk, int s_) {
      const double* fpa = &la[buf][(wm * 64 + lr) * RM_LD + kk * 4 + lk];
      const double* fpb = &lb[buf][(kk * 4 + lk) * KM_LD + wn * 32 + ((F & 512) ? 4 * (lr & 3) + (lr >> 2) : lr)];
#pragma unroll
      for (int i = 0; i < 4; ++i) ga[s_][i] = fpa[i * 16 * RM_LD];
#pragma unroll
      for (int i = 0; i < 2; ++i) gb[s_][i] = fpb[i * 16];
    };
    auto mm8 = [&](int s_) {
      if (F & 32) __builtin_amdgcn_s_setprio(2);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (F & 512) ? __builtin_amdgcn_mfma_f64_16x16x4f64(gb[s_][b], ga[s_][a], acc[a][b], 0, 0, 0)
                                                               : __builtin_amdgcn_mfma_f64_16x16x4f64(ga[s_][a], gb[s_][b], acc[a][b], 0, 0, 0);
      if (F & 32) __builtin_amdgcn_s_setprio(0);
    };
    frag(0, 0, 0);
    for (int k0 = 0; k0 < K; k0 += BK) {
      if (!(F & 256)) {
        if (k0 + BK < K) stage(cur ^ 1);
        if (k0 + 2 * BK < K) load(k0 + 2 * BK);
      }
      frag(cur, 1, 1);
      __builtin_amdgcn_sched_barrier(0);
      mm8(0);
      __builtin_amdgcn_sched_barrier(0);
      if (F & 256) {
        if (k0 + BK < K) stage(cur ^ 1);
        if (k0 + 2 * BK < K) load(k0 + 2 * BK);
        __builtin_amdgcn_sched_barrier(0);
      }
      frag(cur, 2, 0);
      __builtin_amdgcn_sched_barrier(0);
      mm8(1);
      __builtin_amdgcn_sched_barrier(0);
      frag(cur, 3, 1);
      __builtin_amdgcn_sched_barrier(0);
      mm8(0);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      if (k0 + BK < K) frag(cur ^ 1, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      mm8(1);
      __builtin_amdgcn_sched_barrier(0);
      cur ^= 1;
    }
  } else if (F & 64) {
    unsigned i = 0;
    for (int k0 = 0; k0 < K; k0 += BK, ++i) {
      if (k0 + BK < K) {
        wait_ge(1, 8u * i);          // every wave has consumed tile i - 1: its buffer may be overwritten
        stage(cur ^ 1);
        arrive(0);
      }
      if (k0 + 2 * BK < K) load(k0 + 2 * BK);
      if (i) wait_ge(0, 8u * i);     // every wave has staged its part of tile i
      mma(cur);
      arrive(1);
      cur ^= 1;
    }
  } else
  for (int k0 = 0; k0 < K; k0 += BK) {
    if (!(F & 4) && k0 + BK < K) stage(cur ^ 1);
    if (!(F & 2) && k0 + 2 * BK < K) load(k0 + 2 * BK);
    mma(cur);
    if (!(F & 1)) __syncthreads();
    cur ^= 1;
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double* crow = C + (long long)(i0 + wm * 64 + a * 16 + 4 * r + lk) * N + j0 + wn * 32;
#pragma unroll
      for (int b = 0; b < 2; ++b) crow[b * 16 + lr] = acc[a][b][r] + ((F & 6) ? ra[0] + rb[0] : 0.0);
    }
}

// 16 waves per block, block tile 256 x 128 (one block per CU, 4 waves per SIMD), the cross-barrier fragment loop: twice the MFMA work
// per tile prologue / epilogue and per B-tile load of the 128 x 128 kernel.  LDS 2 x (256 x 18 + 16 x 144) doubles = 108 KB (dynamic).
constexpr int BM2 = 256, NT2 = 1024, TILE_A2 = BM2 * RM_LD;
__global__ __launch_bounds__(NT2, 4) void gemm256(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C,
                                                  int n, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) double dyn[];
  double* la0 = dyn;                      // [2][TILE_A2]
  double* lb0 = dyn + 2 * TILE_A2;        // [2][TILE]
  const int tiles_n = N / BN;
  int v = blockIdx.x;
  const int ntiles = (n / BM2) * tiles_n;
  if ((ntiles & 7) == 0) { const int cpx = ntiles >> 3; v = (v & 7) * cpx + (v >> 3); }
  const int ti = v / tiles_n, tj = v - ti * tiles_n, i0 = ti * BM2, j0 = tj * BN;
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), lr = lane & 15, lk = lane >> 4;
  const int wm = w >> 2, wn = w & 3;
  f64x4 acc[4][2];
  for (int a = 0; a < 4; ++a)
    for (int b = 0; b < 2; ++b) acc[a][b] = f64x4{0, 0, 0, 0};
  double ra[4], rb[2];
  const int ar = t >> 2, ak = (t & 3) * 4, bk = t >> 6, bc = (t & 63) * 2;
  const double* pa = A + (long long)(i0 + ar) * K + ak;
  const double* pb = B + (long long)bk * N + j0 + bc;
  auto load = [&](int k0) {
    const f64x2 x0 = *reinterpret_cast<const f64x2*>(pa + k0), x1 = *reinterpret_cast<const f64x2*>(pa + k0 + 2);
    ra[0] = x0.x, ra[1] = x0.y, ra[2] = x1.x, ra[3] = x1.y;
    const f64x2 y0 = *reinterpret_cast<const f64x2*>(pb + (long long)k0 * N);
    rb[0] = y0.x, rb[1] = y0.y;
  };
  auto stage = [&](int buf) {
    double* sa = la0 + buf * TILE_A2 + ar * RM_LD + ak;
    *reinterpret_cast<f64x2*>(sa) = f64x2{ra[0], ra[1]};
    *reinterpret_cast<f64x2*>(sa + 2) = f64x2{ra[2], ra[3]};
    *reinterpret_cast<f64x2*>(lb0 + buf * TILE + bk * KM_LD + bc) = f64x2{rb[0], rb[1]};
  };
  double ga[2][4], gb[2][2];
  auto frag = [&](int buf, int kk, int s_) {
    const double* fpa = la0 + buf * TILE_A2 + (wm * 64 + lr) * RM_LD + kk * 4 + lk;
    const double* fpb = lb0 + buf * TILE + (kk * 4 + lk) * KM_LD + wn * 32 + lr;
#pragma unroll
    for (int i = 0; i < 4; ++i) ga[s_][i] = fpa[i * 16 * RM_LD];
#pragma unroll
    for (int i = 0; i < 2; ++i) gb[s_][i] = fpb[i * 16];
  };
  auto mm8 = [&](int s_) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(ga[s_][a], gb[s_][b], acc[a][b], 0, 0, 0);
  };
  int cur = 0;
  load(0);
  stage(0);
  load(BK);
  __syncthreads();
  frag(0, 0, 0);
  for (int k0 = 0; k0 < K; k0 += BK) {
    frag(cur, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    mm8(0);
    __builtin_amdgcn_sched_barrier(0);
    if (k0 + BK < K) stage(cur ^ 1);
    if (k0 + 2 * BK < K) load(k0 + 2 * BK);
    __builtin_amdgcn_sched_barrier(0);
    frag(cur, 2, 0);
    __builtin_amdgcn_sched_barrier(0);
    mm8(1);
    __builtin_amdgcn_sched_barrier(0);
    frag(cur, 3, 1);
    __builtin_amdgcn_sched_barrier(0);
    mm8(0);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    if (k0 + BK < K) frag(cur ^ 1, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    mm8(1);
    __builtin_amdgcn_sched_barrier(0);
    cur ^= 1;
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double* crow = C + (long long)(i0 + wm * 64 + a * 16 + 4 * r + lk) * N + j0 + wn * 32;
#pragma unroll
      for (int b = 0; b < 2; ++b) crow[b * 16 + lr] = acc[a][b][r];
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
  hipFuncSetAttribute((const void*)gemm256, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * (2 * TILE_A2 + 2 * TILE)));
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const int blocks = (n / BM) * (N / BN);
  const char* names[] = {"baseline (stage-first)", "no barrier", "no global loads", "no ds_write", "no loads, no ds_write",
                         "no ds_read", "no loads/ds_write/ds_read (MFMA + barrier)", "MFMA only", "no barrier, no loads",
                         "setprio 3 around MFMAs", "no barrier + no ds_write + no loads", "LDS-counter sync instead of s_barrier", "fragment prefetch carried across the barrier", "same, stage + load behind the first 8 MFMAs", "same + setprio", "same, transposed accumulators (SWAP)", "16 waves, 256 x 128 tile, one block per CU, same loop"};
  for (int pass = 0; pass < 2; ++pass)     // pass 0 warms the device up (clocks): only pass 1 is printed
  for (int variant = 0; variant < 17; ++variant) {
    auto run = [&] {
#define L(F) hipLaunchKernelGGL(gemm<F>, dim3(blocks), dim3(NT), 0, 0, A, B, C, n, N, K)
      switch (variant) {
        case 0: L(0); break;
        case 1: L(1); break;
        case 2: L(2); break;
        case 3: L(4); break;
        case 4: L(6); break;
        case 5: L(8); break;
        case 6: L(14); break;
        case 7: L(15); break;
        case 8: L(3); break;
        case 9: L(32); break;
        case 10: L(7); break;
        case 11: L(64); break;
        case 12: L(128); break;
        case 13: L(384); break;
        case 14: L(416); break;
        case 15: L(896); break;
        case 16: hipLaunchKernelGGL(gemm256, dim3((n / BM2) * (N / BN)), dim3(NT2), sizeof(double) * (2 * TILE_A2 + 2 * TILE), 0, A, B, C, n, N, K); break;
      }
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
    std::vector<double> c(1);
    hipMemcpy(c.data(), C + 12345 * (size_t)N + 100, sizeof(double), hipMemcpyDeviceToHost);
    double ref = 0;
    for (int k = 0; k < K; ++k) ref += h[(size_t)12345 * K + k] * h[(size_t)k * N + 100];
    if (pass) printf("%-48s %.3f ms  %.1f TFLOP/s   check %.2e\n", names[variant], ms, 2.0 * n * N * K / ms / 1e9, c[0] - ref);
  }
  return 0;
}
