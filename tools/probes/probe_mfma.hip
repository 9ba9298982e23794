// probe_mfma.hip -- prints the lane/register -> (row, col) map of v_mfma_f64_16x16x4_f64 on the device it runs on.
// D = A * B with A[i][k] = 1 + i + 100*k (lane l holds A[l&15][l>>4]) and B[k][j] = (k == 0) * (1 + j) ... the host
// decodes each output value back to (i, j) and reports which formula matches.  Diagnostic only (not part of the .so).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ void probe(double* out) {
  const int l = threadIdx.x;
  const int i = l & 15, k = l >> 4;
  const double a = (k == 0) ? (double)(1 + i) : 0.0;       // A[i][0] = 1+i
  const double b = (k == 0) ? (double)(1000 * (1 + (l & 15))) : 0.0;  // B[0][j] = 1000*(1+j)
  f64x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];  // D[i][j] = (1+i) * 1000 * (1+j)
}
int main() {
  double* d;
  double h[256];
  if (hipMalloc(&d, sizeof h) != hipSuccess) { printf("no device\n"); return 1; }
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  int okA = 1, okB = 1;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const long v = (long)h[l * 4 + r];
      const int j = (int)(v / 1000) / 1;  // (1+i)*(1+j)*1000
      (void)j;
      const int colA = l & 15, rowA = (l >> 4) + 4 * r;      // guide's f64 map
      const int colB = l & 15, rowB = (l >> 4) * 4 + r;      // f32-style map
      if (v != (long)(1 + rowA) * 1000 * (1 + colA)) okA = 0;
      if (v != (long)(1 + rowB) * 1000 * (1 + colB)) okB = 0;
    }
  printf("mfma_f64_16x16x4 D map: row=(lane>>4)+4*reg -> %s ; row=(lane>>4)*4+reg -> %s\n", okA ? "MATCH" : "no", okB ? "MATCH" : "no");
  for (int l = 0; l < 64; l += 16) printf("lane %2d: %g %g %g %g\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  return 0;
}
