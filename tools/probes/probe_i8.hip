// probe_i8.hip -- round 5, VERDICT r4 item 9: bare issue rate of the gfx950 int8 matrix instruction (v_mfma_i32_16x16x64_i8,
// operands in registers, no memory), at 1 / 2 / 4 waves per SIMD, next to v_mfma_f64_16x16x4_f64 in the same harness.
// Together with tools/probes/i8_split_error.py (how many int8 slice products one FP64 product needs for 1e-9) this bounds what an
// error-bounded int8 split of the forward contraction could gain.   hipcc --offload-arch=gfx950 -O3 probe_i8.hip -o probe_i8
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void i8_kernel(int iters, int* out) {
  v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)blockIdx.x, 7};
  v4i acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = v4i{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i], 0, 0, 0);
  }
  int s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 0x7fffffff) out[0] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void f64_kernel(int iters, double* out) {
  double a = threadIdx.x * 1e-3, b = blockIdx.x * 1e-3;
  v4d acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = v4d{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 1.2345) out[0] = s;
}

int main() {
  int* d;
  hipMalloc(&d, 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const int iters = 20000;
  for (int blocks_per_cu : {1, 2, 4}) {            // 256-thread blocks = 4 waves = 1 wave per SIMD each
    const int grid = 256 * blocks_per_cu;
    for (int kind = 0; kind < 2; ++kind) {
      for (int rep = 0; rep < 2; ++rep) {          // (second pass: past the clock ramp)
        hipEventRecord(e0);
        if (kind == 0) hipLaunchKernelGGL((i8_kernel<8>), dim3(grid), dim3(256), 0, 0, iters, d);
        else hipLaunchKernelGGL((f64_kernel<8>), dim3(grid), dim3(256), 0, 0, iters, (double*)d);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double mfmas = (double)grid * 4 * iters * 8;
        const double ops = mfmas * (kind == 0 ? 2.0 * 16 * 16 * 64 : 2.0 * 16 * 16 * 4);
        if (rep == 1)
          std::printf("%s  %d wave(s) per SIMD: %.3f ms  %.1f T%s/s\n", kind == 0 ? "v_mfma_i32_16x16x64_i8 " : "v_mfma_f64_16x16x4_f64 ",
                      blocks_per_cu, ms, ops / ms / 1e9, kind == 0 ? "OP" : "FLOP");
      }
    }
  }
  return 0;
}
