// Issue rate of v_pk_fma_f32 on gfx950 in the forms the compositors use: plain register pairs, with the second operand's
// low half broadcast to both halves (op_sel_hi:[1,0,1], what `acc2 = fma(c2, splat(w), acc2)` compiles to), and scalar
// v_fma_f32 for reference.  256 workgroups x 1024 threads = 4 wavefronts per SIMD on every CU; chain-free accumulators.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
__global__ void __launch_bounds__(1024) k(float *out, int iters, int mode) {
  f2 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f2{threadIdx.x * 1e-3f + i, 1.f};
  f2 c = {1.0001f, 0.9999f}, w = {0.5f, 0.25f};
  asm volatile("" : "+v"(c), "+v"(w));
  if (mode == 0) {
    for (int it = 0; it < iters; ++it) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(c), "v"(w));
      REP16(X)
#undef X
    }
  } else if (mode == 1) {
    for (int it = 0; it < iters; ++it) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[i]) : "v"(c), "v"(w));
      REP16(X)
#undef X
    }
  } else if (mode == 2) {
    for (int it = 0; it < iters; ++it) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].x) : "v"(c.x), "v"(w.x));
      REP16(X)
#undef X
    }
  } else if (mode == 3) {   // packed multiply and add
    for (int it = 0; it < iters; ++it) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(acc[i]) : "v"(c));
      REP16(X)
#undef X
    }
  } else {                  // broadcast on the FIRST operand instead
    for (int it = 0; it < iters; ++it) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "v"(c), "v"(w));
      REP16(X)
#undef X
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].y;
  out[blockIdx.x * 1024 + threadIdx.x] = s;
}
int main() {
  float *out;
  if (hipMalloc(&out, 256 * 4 * 1024 * 4) != hipSuccess) return 1;
  const int iters = 4000;
  const char *names[5] = {"v_pk_fma_f32 plain pairs", "v_pk_fma_f32 op_sel_hi:[1,0,1]", "v_fma_f32", "v_pk_mul_f32", "v_pk_fma_f32 op_sel_hi:[0,1,1]"};
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int m = 0; m < 5; ++m) {
    hipLaunchKernelGGL(k, dim3(1024), dim3(1024), 0, 0, out, 10, m);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(1024), dim3(1024), 0, 0, out, iters, m);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // 1024 blocks x 16 waves = 16384 waves over 1024 SIMDs = 16 waves per SIMD in 4 rounds of 4
    const double instr_per_simd = 16.0 * iters * 16;
    printf("%-34s %.3f ms  -> %.2f ns per wave-instruction per SIMD (4.0 cycles = %.2f ns at 2.4 GHz)\n", names[m], ms, ms * 1e6 / instr_per_simd, 4 / 2.4);
  }
  return 0;
}
