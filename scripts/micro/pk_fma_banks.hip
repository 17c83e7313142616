// Does v_pk_fma_f32 slow down when two of its 64-bit register operands start in the same register bank (index mod 4)?
// Explicit registers: accumulators v[16:47], multiplicand pair c, weight pair w at chosen indices.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define FMA(a, c, w) "v_pk_fma_f32 v[" #a ":" #a "+1], v[" #c ":" #c "+1], v[" #w ":" #w "+1], v[" #a ":" #a "+1] op_sel_hi:[1,0,1]\n\t"
#define BODY(c, w) FMA(16, c, w) FMA(18, c, w) FMA(20, c, w) FMA(22, c, w) FMA(24, c, w) FMA(26, c, w) FMA(28, c, w) FMA(30, c, w) \
                   FMA(32, c, w) FMA(34, c, w) FMA(36, c, w) FMA(38, c, w) FMA(40, c, w) FMA(42, c, w) FMA(44, c, w) FMA(46, c, w)
#define CLOB "v2","v3","v4","v5","v6","v7","v8","v9","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47"
__global__ void __launch_bounds__(1024) k(float *out, int iters, int mode) {
  asm volatile("v_mov_b32 v2, 1.0\n\tv_mov_b32 v3, 1.0\n\tv_mov_b32 v4, 0.5\n\tv_mov_b32 v5, 0.5\n\tv_mov_b32 v6, 0.5\n\tv_mov_b32 v7, 0.5\n\tv_mov_b32 v8, 1.0\n\tv_mov_b32 v9, 1.0" ::: CLOB);
  for (int it = 0; it < iters; ++it) {
    if (mode == 0) asm volatile(BODY(2, 4) ::: CLOB);        // c bank 2, w bank 0, acc banks 0/2
    else if (mode == 1) asm volatile(BODY(2, 6) ::: CLOB);   // c bank 2, w bank 2
    else if (mode == 2) asm volatile(BODY(8, 4) ::: CLOB);   // c bank 0, w bank 0
    else asm volatile(BODY(8, 6) ::: CLOB);                  // c bank 0, w bank 2
  }
  float s;
  asm volatile("v_add_f32 %0, v16, v46" : "=v"(s) :: CLOB);
  out[blockIdx.x * 1024 + threadIdx.x] = s;
}
int main() {
  float *out;
  if (hipMalloc(&out, 1024 * 1024 * 4) != hipSuccess) return 1;
  const int iters = 4000;
  const char *names[4] = {"c@v2 w@v4", "c@v2 w@v6", "c@v8 w@v4", "c@v8 w@v6"};
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int m = 0; m < 4; ++m) {
    hipLaunchKernelGGL(k, dim3(1024), dim3(1024), 0, 0, out, 10, m);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(1024), dim3(1024), 0, 0, out, iters, m);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-18s %.3f ms -> %.2f ns per wave-instruction per SIMD\n", names[m], ms, ms * 1e6 / (16.0 * iters * 16));
  }
  return 0;
}
