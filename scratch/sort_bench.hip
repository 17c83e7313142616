// Micro-benchmark: rocPRIM radix_sort_pairs on 1M (u32,u32) pairs under several Onesweep configurations.
#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdio>
#include <vector>
#include <rocprim/device/device_radix_sort.hpp>

template <class Cfg>
float run(const char *name, size_t n, unsigned end_bit) {
  uint32_t *k_in, *k_out, *v_in, *v_out;
  hipMalloc(&k_in, n * 4); hipMalloc(&k_out, n * 4); hipMalloc(&v_in, n * 4); hipMalloc(&v_out, n * 4);
  std::vector<uint32_t> h(n);
  uint32_t s = 12345;
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; float f = 0.5f + (s >> 8) * (20.f / 16777216.f); memcpy(&h[i], &f, 4); if (end_bit < 32) h[i] = s >> (32 - end_bit); }
  hipMemcpy(k_in, h.data(), n * 4, hipMemcpyHostToDevice);
  hipMemcpy(v_in, h.data(), n * 4, hipMemcpyHostToDevice);
  size_t tb = 0;
  rocprim::radix_sort_pairs<Cfg>(nullptr, tb, k_in, k_out, v_in, v_out, n, 0u, end_bit);
  void *temp; hipMalloc(&temp, tb);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) rocprim::radix_sort_pairs<Cfg>(temp, tb, k_in, k_out, v_in, v_out, n, 0u, end_bit);
  hipEventRecord(a);
  for (int i = 0; i < 20; ++i) rocprim::radix_sort_pairs<Cfg>(temp, tb, k_in, k_out, v_in, v_out, n, 0u, end_bit);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-28s n=%zu bits=%u  %.1f us\n", name, n, end_bit, ms * 1000.f / 20);
  hipFree(k_in); hipFree(k_out); hipFree(v_in); hipFree(v_out); hipFree(temp);
  return ms;
}
using namespace rocprim;
template <unsigned BS, unsigned IPT, unsigned BITS>
using OS = radix_sort_config<default_config, default_config,
                             radix_sort_onesweep_config<kernel_config<256, 12>, kernel_config<BS, IPT>, BITS>, 32768>;
int main() {
  for (size_t n : {(size_t)1000000, (size_t)200000}) {
    run<default_config>("default", n, 32);
    run<radix_sort_config<default_config, default_config, default_config, 32768>>("onesweep default", n, 32);
    run<OS<256, 4, 8>>("os 256x4 b8", n, 32);
    run<OS<256, 8, 8>>("os 256x8 b8", n, 32);
    run<OS<256, 16, 8>>("os 256x16 b8", n, 32);
    run<OS<256, 12, 7>>("os 256x12 b7", n, 32);
    run<OS<256, 16, 6>>("os 256x16 b6", n, 32);
    run<OS<128, 8, 8>>("os 128x8 b8", n, 32);
  }
  for (size_t n : {(size_t)11670455, (size_t)2900000}) {
    run<default_config>("default", n, 13);
    run<OS<256, 16, 7>>("os 256x16 b7", n, 13);
    run<OS<256, 24, 7>>("os 256x24 b7", n, 13);
    run<OS<256, 16, 8>>("os 256x16 b8", n, 13);
  }
  return 0;
}
