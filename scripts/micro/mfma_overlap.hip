// Do float32 matrix instructions and vector instructions of DIFFERENT wavefronts on one SIMD overlap on gfx950?
// Workgroups of 512 threads = 8 wavefronts = 2 per SIMD.  mode 0: all wavefronts run a chain-free v_fma loop;
// mode 1: all run a v_mfma_f32_16x16x4_f32 loop; mode 2: one wavefront per SIMD runs each;
// mode 3: every wavefront interleaves both in its own instruction stream (1 mfma : 7 fma).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
#ifdef USE_BF16
#define MFMA(acc) __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc, 0, 0, 0)
#else
#define MFMA(acc) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0)
#endif
__device__ __forceinline__ void valu_block(float (&x)[8], float a, float b) {
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i], a, b);
}
__global__ void __launch_bounds__(512) k(float *out, int iters, int mode, int selbit) {
  const int wave = threadIdx.x >> 6;
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i;
  f4v acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f4v{0.f, 0.f, 0.f, 0.f};
  const float a = 1.0001f, b = 0.5f;
  bf8v ab, bb;
  for (int i = 0; i < 8; ++i) ab[i] = (__bf16)(0.5f + i), bb[i] = (__bf16)(1.f + threadIdx.x);
  const bool do_valu = mode == 0 || (mode == 2 && ((wave >> selbit) & 1)) || mode == 3;   // waves 0-3 and 4-7 land on SIMDs 0-3 each
  const bool do_mfma = mode == 1 || (mode == 2 && !((wave >> selbit) & 1)) || mode == 3;
  if (mode == 3) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[j] = MFMA(acc[j]);
        valu_block(x, a, b);   // 8 fma after each mfma (32 cycles of issue = one mfma's duration)
      }
    }
  } else {
    if (do_valu)
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) valu_block(x, a, b);
      }
    if (do_mfma)
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = MFMA(acc[j]);
      }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += x[i];
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main() {
  float *out;
  if (hipMalloc(&out, 256 * 512 * 4) != hipSuccess) return 1;
  const int iters = 20000;
  const char *names[4] = {"fma only (32 fma / iteration / wave)", "mfma only (4 mfma / iteration / wave)",
                          "half the waves each", "interleaved in every wave (4 mfma + 32 fma)"};
  for (int m = 0; m < 6; ++m) {
    const int mode = m < 2 ? m : m < 5 ? 2 : 3, selbit = m < 2 ? 0 : m - 2;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, 100, mode, selbit);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, iters, mode, selbit);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // cycles per iteration per SIMD at 2.4 GHz
    printf("sel %d %-48s %8.3f ms  = %7.1f cycles per iteration per SIMD\n", selbit, names[mode], ms, ms * 1e-3 * 2.4e9 / iters);
  }
  return 0;
}
