#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
  unsigned l = threadIdx.x;
  unsigned a = l, b = 1000 + l;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[l] = r[0]; out[64 + l] = r[1];
  auto s = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[128 + l] = s[0]; out[192 + l] = s[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 256 * 4); k<<<1, 64>>>(d); unsigned h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
  for (int t = 0; t < 4; ++t) { printf("out%d:", t); for (int l = 0; l < 64; l += 8) printf(" [%d]=%u", l, h[64 * t + l]); printf("\n"); }
  return 0;
}
