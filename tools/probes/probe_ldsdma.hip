// Probe: semantics of global_load_lds_dwordx4 on gfx950 (where does lane l's 16 bytes land; does vmcnt cover the LDS write)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const double* g, double* out, int half) {
  __shared__ __attribute__((aligned(16))) double buf[2][128];
  const int l = threadIdx.x;
  for (int i = l; i < 256; i += 64) (&buf[0][0])[i] = -1.0;
  __syncthreads();
  const int src = (l * 7) & 63;  // permuted source: lane l fetches doubles 2*src, 2*src+1
  if (!half || l < 32)
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(g + 2 * src),
                                     (void __attribute__((address_space(3)))*)&buf[1][0], 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int i = l; i < 256; i += 64) out[i] = (&buf[0][0])[i];
}
int main() {
  double h[128], *g, *o, r[256];
  for (int i = 0; i < 128; ++i) h[i] = i;
  hipMalloc(&g, sizeof h), hipMalloc(&o, sizeof r);
  hipMemcpy(g, h, sizeof h, hipMemcpyHostToDevice);
  for (int half = 0; half < 2; ++half) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, g, o, half);
    hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int i = 0; i < 128; ++i) ok &= (r[i] == -1.0);
    for (int l = 0; l < 64; ++l) {
      const int src = (l * 7) & 63;
      const bool act = !half || l < 32;
      ok &= act ? (r[128 + 2 * l] == 2 * src && r[128 + 2 * l + 1] == 2 * src + 1) : (r[128 + 2 * l] == -1.0);
    }
    printf("half=%d lane-linear placement %s  (lane1 -> %g %g)\n", half, ok ? "OK" : "MISMATCH", r[130], r[131]);
  }
  return 0;
}
