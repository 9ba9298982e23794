// probe_hwid.hip -- which SIMD / CU / XCC do the 8 waves of 512-thread blocks land on (gfx950)?  Diagnostic only.
// Prints, for a few blocks, per wave: HW_REG_HW_ID raw, simd (bits 5:4), cu (11:8), sh (12), se (15:13), XCC_ID.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(512, 4) void k(unsigned* out) {
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));       // HW_REG_HW_ID, offset 0, size 32
    unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));     // HW_REG_XCC_ID
    out[(blockIdx.x * 8 + w) * 2 + 0] = hw;
    out[(blockIdx.x * 8 + w) * 2 + 1] = xcc;
  }
  // keep the block resident for a while so that two blocks per CU overlap
  long long t0 = clock64();
  while (clock64() - t0 < 200000) {}
}
int main() {
  const int nb = 1024;
  unsigned* d;
  hipMalloc(&d, nb * 8 * 2 * sizeof(unsigned));
  hipLaunchKernelGGL(k, dim3(nb), dim3(512), 0, 0, d);
  std::vector<unsigned> h(nb * 16);
  hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
  for (int b : {0, 1, 8, 256, 257, 512, 1023}) {
    printf("block %4d:", b);
    for (int w = 0; w < 8; ++w) {
      unsigned hw = h[(b * 8 + w) * 2], xcc = h[(b * 8 + w) * 2 + 1];
      printf("  w%d[simd %u cu %u sh %u se %u xcc %u wv %u]", w, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, xcc & 15, hw & 15);
    }
    printf("\n");
  }
  // statistics: are waves w and w+4 always on the same SIMD, and the 4 pairs on 4 distinct SIMDs?
  int same = 0, distinct = 0;
  for (int b = 0; b < nb; ++b) {
    bool s = true;
    unsigned mask = 0;
    for (int w = 0; w < 4; ++w) {
      unsigned a = (h[(b * 8 + w) * 2] >> 4) & 3, c = (h[(b * 8 + w + 4) * 2] >> 4) & 3;
      s = s && a == c;
      mask |= 1u << a;
    }
    same += s, distinct += mask == 15u;
  }
  printf("blocks with (w, w+4) on one SIMD: %d / %d;  four pairs on four SIMDs: %d / %d\n", same, nb, distinct, nb);
  return 0;
}
