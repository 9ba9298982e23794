// probe_clock.hip -- what shader clock does a tiny kernel (one wave, nothing else on the device) actually run at?
// s_sleep 127 pauses a wave for 127 x 64 shader cycles; wall_clock64() counts at a constant 100 MHz.
// Build: hipcc -O2 --offload-arch=gfx950 probe_clock.hip -o build/probe_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void sleeper(long long* out, int reps) {
  const long long t0 = wall_clock64();
  const long long c0 = clock64();
  for (int i = 0; i < reps; ++i) __builtin_amdgcn_s_sleep(127);
  const long long c1 = clock64();
  const long long t1 = wall_clock64();
  if (threadIdx.x == 0) out[0] = t1 - t0, out[1] = c1 - c0;
}
__global__ void fma_chain(double* out, long long* tk, int reps) {
  double x = 1.0 + threadIdx.x * 1e-9;
  const long long t0 = wall_clock64();
  for (int i = 0; i < reps; ++i) {
#pragma unroll
    for (int u = 0; u < 64; ++u) x = fma(x, 1.0000001, 1e-9);
  }
  const long long t1 = wall_clock64();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) tk[0] = t1 - t0;
}
__global__ void burn(double* out, int reps) {
  double x = threadIdx.x;
  for (int i = 0; i < reps; ++i) x = fma(x, 1.0000001, 1e-9);
  out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

int main() {
  long long* d;
  double* o;
  hipMalloc(&d, 64), hipMalloc(&o, sizeof(double) * 1024 * 256);
  long long h[2];
  for (int pass = 0; pass < 3; ++pass) {
    if (pass == 2) {   // the whole chip busy for ~0.5 s first
      for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(burn, dim3(1024), dim3(256), 0, 0, o, 2000000);
      hipDeviceSynchronize();
    }
    for (int k = 0; k < 3; ++k) {
      hipLaunchKernelGGL(sleeper, dim3(1), dim3(64), 0, 0, d, 200);
      hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
      const double ns = h[0] * 10.0, cyc = 200.0 * 127 * 64;
      printf("pass %d: s_sleep: %.0f cycles in %.0f ns -> %.2f GHz; clock64 ticks per ns %.3f\n", pass, cyc, ns, cyc / ns, h[1] / ns);
    }
    hipLaunchKernelGGL(fma_chain, dim3(1), dim3(64), 0, 0, o, d, 1000);
    hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("pass %d: dependent v_fma_f64: %.2f ns each\n", pass, h[0] * 10.0 / 64000.0);
  }
  return 0;
}
