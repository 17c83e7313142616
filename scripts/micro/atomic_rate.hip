// Float-atomic delivery rate on gfx950 for the access shapes the backward compositors produce.
// Each wavefront issues ITER fire-and-forget global_atomic_add_f32 instructions; lane l of instruction i adds to
// row r(i, l / SEG) * LD + (l % SEG) of a [ROWS, LD] float table, rows drawn pseudo-randomly per segment.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__device__ __forceinline__ void atomic_add_noret(float *addr, float v) {
  asm volatile("global_atomic_add_f32 %0, %1, off" : : "v"(addr), "v"(v) : "memory");
}
template <int SEG, int ACTIVE>
__global__ void __launch_bounds__(64) k(float *tab, int rows, int ld, int iters, int local) {
  const int lane = threadIdx.x;
  unsigned s = blockIdx.x * 2654435761u + 12345u;
  for (int i = 0; i < iters; ++i) {
    s = s * 1664525u + 1013904223u;
    unsigned h = (s >> 8) + (lane / SEG) * 7919u;
    // local: rows near a per-block base (Gaussians of neighbouring tiles), else anywhere
    unsigned row = local ? ((blockIdx.x * 37u) + (h % 2048u)) % (unsigned)rows : h % (unsigned)rows;
    if (lane < ACTIVE) atomic_add_noret(tab + (size_t)row * ld + (lane % SEG), 1.0f);
  }
}
template <int SEG, int ACTIVE>
void run(const char *name, float *tab, int rows, int ld, int local) {
  const int blocks = 32768, iters = 200;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<SEG, ACTIVE>), dim3(blocks), dim3(64), 0, 0, tab, rows, ld, 10, local);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<SEG, ACTIVE>), dim3(blocks), dim3(64), 0, 0, tab, rows, ld, iters, local);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double instr = (double)blocks * iters, adds = instr * ACTIVE;
  printf("%-44s ld=%3d local=%d  %7.3f ms  %7.1f G adds/s  %6.2f G instr/s  %6.2f G segments/s\n", name, ld, local, ms,
         adds / ms * 1e-6, instr / ms * 1e-6, instr * ((ACTIVE + SEG - 1) / SEG) / ms * 1e-6);
}
int main() {
  const int rows = 1000000;
  float *tab;
  hipMalloc(&tab, (size_t)rows * 64 * 4);
  hipMemset(tab, 0, (size_t)rows * 64 * 4);
  for (int local = 0; local < 2; ++local) {
    run<64, 64>("64 contiguous floats, 1 row", tab, rows, 64, local);
    run<64, 36>("36 contiguous floats, 1 row", tab, rows, 36, local);
    run<64, 32>("32 contiguous floats, 1 row (aligned)", tab, rows, 32, local);
    run<64, 44>("44 contiguous floats, 1 row", tab, rows, 44, local);
    run<16, 64>("4 rows x 16 floats (aligned)", tab, rows, 32, local);
    run<16, 64>("4 rows x 16 floats (ld 36: straddling)", tab, rows, 36, local);
    run<8, 64>("8 rows x 8 floats", tab, rows, 8, local);
    run<8, 8>("1 row x 8 floats", tab, rows, 8, local);
    run<4, 64>("16 rows x 4 floats", tab, rows, 4, local);
    run<1, 64>("64 scattered floats", tab, rows, 1, local);
  }
  return 0;
}
