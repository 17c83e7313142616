#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../pogs_amd/csrc/common.h"
namespace pr { void set_hip_error(hipError_t, const char*) {} }
__global__ void k(float* out) {
  int l = threadIdx.x;
  // v_i[l] = 1000*i + l
  float v0 = l, v1 = 1000 + l, v2 = 2000 + l, v3 = 3000 + l;
  out[l] = pr::swap32_add(v0, v1);
  out[64 + l] = pr::swap16_add(v0, v1);
  out[128 + l] = pr::quad_reduce(v0, v1, v2, v3);
  out[192 + l] = pr::row_sum_to_lane15(v0);
}
int main() {
  float* d; hipMalloc(&d, 256 * 4); k<<<1, 64>>>(d); float h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
  const char* names[4] = {"swap32_add", "swap16_add", "quad_reduce", "row_sum"};
  for (int t = 0; t < 4; ++t) { printf("%s:", names[t]); for (int l = 0; l < 64; ++l) if (t < 2 ? (l % 16 == 0 || l%16==1) : (l % 16 == 15)) printf(" [%d]=%g", l, h[64 * t + l]); printf("\n"); }
  return 0;
}
