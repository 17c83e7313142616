// Micro-benchmark: rocPRIM radix_sort_pairs with 13 significant key bits, u32 vs u16 keys, u32 values.
#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdio>
#include <vector>
#include <rocprim/device/device_radix_sort.hpp>
template <class K>
void run(const char *name, size_t n, unsigned bits) {
  K *k_in, *k_out; uint32_t *v_in, *v_out;
  hipMalloc(&k_in, n * sizeof(K)); hipMalloc(&k_out, n * sizeof(K)); hipMalloc(&v_in, n * 4); hipMalloc(&v_out, n * 4);
  std::vector<K> h(n);
  uint32_t s = 12345;
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (K)((s >> 8) % 8160u); }
  hipMemcpy(k_in, h.data(), n * sizeof(K), hipMemcpyHostToDevice);
  size_t tb = 0;
  rocprim::radix_sort_pairs(nullptr, tb, k_in, k_out, v_in, v_out, n, 0u, bits);
  void *temp; hipMalloc(&temp, tb);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) rocprim::radix_sort_pairs(temp, tb, k_in, k_out, v_in, v_out, n, 0u, bits);
  hipEventRecord(a);
  for (int i = 0; i < 20; ++i) rocprim::radix_sort_pairs(temp, tb, k_in, k_out, v_in, v_out, n, 0u, bits);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-10s n=%zu bits=%u  %.1f us\n", name, n, bits, ms * 1000.f / 20);
}
int main() {
  for (size_t n : {(size_t)6391220, (size_t)1500000}) {
    run<uint32_t>("u32", n, 13);
    run<uint16_t>("u16", n, 13);
    run<uint16_t>("u16/16", n, 16);
  }
  return 0;
}
